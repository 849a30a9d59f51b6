#!/bin/bash
# round-5 GPU lease: A/B of the two arithmetics on ONE box (per-layer isolated timings, frame latency, pipelined value) + CU masks
mkdir -p gpurun_out/r5f
export PYTHONPATH=/root/repo:$PYTHONPATH
timeout 300 python -m pytest tests/test_gpu_plan.py tests/test_gpu_dense_conv.py -x -q -m gpu 2>&1 | tail -5
for p in fp32 bf16x3 fp32; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fast-mode --no-h2d --windows 9 --precision $p > gpurun_out/r5f/bench_$p.json 2> gpurun_out/r5f/bench_$p.err
  python - "$p" <<'PY'
import json, sys
p = sys.argv[1]
d = json.loads(open(f"gpurun_out/r5f/bench_{p}.json").read().strip().splitlines()[-1])
print(p, "value %.0f single_ms %.4f dense_us %.2f sparse_us %.1f backbone %.1f prebuilt %.1f" % (d["value"], d["single_frame_ms"], d["roofline_dense"]["avg_us"], d["stages"]["sparse_conv_us"], d["stages"]["backbone_us_with_voxelizer_and_rulebook_build"], d["stages"]["backbone_us_rulebooks_prebuilt"]))
print("   ", [l["t_avg_us"] for l in d["stages"]["layers"]])
PY
done
./tools/mb_cumask > gpurun_out/r5f/cumask_probe.txt 2>&1; cat gpurun_out/r5f/cumask_probe.txt
timeout 600 python tools/mb_cu_mask_pipeline.py > gpurun_out/r5f/cu_mask_pipeline.txt 2>&1; tail -8 gpurun_out/r5f/cu_mask_pipeline.txt
