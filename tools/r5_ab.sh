#!/bin/bash
# round-5 GPU lease: A/B of the two arithmetics on ONE box (per-layer isolated timings, frame latency, pipelined value)
mkdir -p gpurun_out/r5e
export PYTHONPATH=/root/repo:$PYTHONPATH
for p in fp32 bf16x3 fp32 bf16x3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast-mode --no-h2d --windows 9 --precision $p > gpurun_out/r5e/bench_$p.json 2> gpurun_out/r5e/bench_$p.err
  python - "$p" <<'PY'
import json, sys
p = sys.argv[1]
d = json.loads(open(f"gpurun_out/r5e/bench_{p}.json").read().strip().splitlines()[-1])
print(p, "value %.0f single_ms %.4f dense_us %.2f sparse_us %.1f backbone %.1f prebuilt %.1f" % (d["value"], d["single_frame_ms"], d["roofline_dense"]["avg_us"], d["stages"]["sparse_conv_us"], d["stages"]["backbone_us_with_voxelizer_and_rulebook_build"], d["stages"]["backbone_us_rulebooks_prebuilt"]))
print("   ", [l["t_avg_us"] for l in d["stages"]["layers"]])
PY
done
timeout 300 python bench.py --mode plumbing --steps 300 --warmup 30 > gpurun_out/r5e/bench_plumbing.json 2> gpurun_out/r5e/bench_plumbing.err; tail -c 600 gpurun_out/r5e/bench_plumbing.err; cut -c1-1500 gpurun_out/r5e/bench_plumbing.json
