# round-4 GPU pass N: sparse conv tests, pipelined bench + overlap trace after the hint-sized ring grid
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_sparse_conv.py tests/test_gpu_plan.py tests/test_gpu_second.py tests/test_gpu_configs.py tests/test_gpu_train_plan.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4n_tests.txt
for rep in 1 2; do for st in 20 300; do
python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $st ->', round(d['value'],1), 'p10', round(d['value_p10'],1), 'p90', round(d['value_p90'],1), 'single', round(d['single_frame_ms'],4), d['config']['pipeline_tuning'])" >> gpurun_out/r4n_tests.txt
done; done
rm -rf /tmp/prof_o
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_o -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-h2d > /dev/null 2> /tmp/prof_o.err
python tools/trace_overlap.py $(find /tmp/prof_o -name "*kernel_trace.csv" | head -1) > gpurun_out/r4n_trace_overlap.txt 2>&1
cat gpurun_out/r4n_tests.txt; head -12 gpurun_out/r4n_trace_overlap.txt
