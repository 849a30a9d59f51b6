#!/bin/bash
# One GPU-box visit: smoke, parity tests, bench, rocprof kernel stats.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== smoke" | tee gpurun_out/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/summary.txt
tail -3 gpurun_out/smoke.log | tee -a gpurun_out/summary.txt
echo "== pytest -m gpu" | tee -a gpurun_out/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_EXTRA:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/summary.txt
tail -40 gpurun_out/pytest_gpu.log | tee -a gpurun_out/summary.txt
echo "== bench" | tee -a gpurun_out/summary.txt
timeout 900 python bench.py --steps ${BENCH_STEPS:-30} --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" | tee -a gpurun_out/summary.txt
tail -c 6000 gpurun_out/bench.json | tee -a gpurun_out/summary.txt; tail -5 gpurun_out/bench.err | tee -a gpurun_out/summary.txt
if [ "${DO_PROF:-1}" = "1" ]; then
  echo "== rocprofv3 kernel stats" | tee -a gpurun_out/summary.txt
  export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "rocprof rc=$?" | tee -a gpurun_out/summary.txt
  find gpurun_out/prof -name "*kernel_stats*" | head -3 | tee -a gpurun_out/summary.txt
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | tee -a gpurun_out/summary.txt
  # keep the trace small: drop the raw per-dispatch csv if huge
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
fi
echo "== done" | tee -a gpurun_out/summary.txt
