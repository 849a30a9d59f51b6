#!/bin/bash
mkdir -p gpurun_out/r5j
export PYTHONPATH=/root/repo:$PYTHONPATH
timeout 1200 python -m pytest tests/test_gpu_plan.py tests/test_gpu_proposal.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r5j/pytest.txt; tail -6 gpurun_out/r5j/pytest.txt
run() { # name, env, args
  env $2 timeout 400 python bench.py --no-cpu-baseline --no-fast-mode --no-h2d --windows 9 --steps 20 --warmup 5 $3 > gpurun_out/r5j/$1.json 2> gpurun_out/r5j/$1.err
  python - "$1" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r5j/{n}.json").read().strip().splitlines()[-1])
    print(n, "value %.0f single_ms %.4f backbone %.1f prebuilt %.1f" % (d["value"], d["single_frame_ms"],
        d["stages"]["backbone_us_with_voxelizer_and_rulebook_build"], d["stages"]["backbone_us_rulebooks_prebuilt"]))
except Exception as e:
    print(n, "failed", e); print(open(f"gpurun_out/r5j/{n}.err").read()[-1500:])
PY
}
run k_fp32_on "V3D_PRESPLIT=1" ""
run k_fp32_off "V3D_PRESPLIT=0" ""
run k_bf16_on "V3D_PRESPLIT=1" "--precision bf16x3"
run k_bf16_off "V3D_PRESPLIT=0" "--precision bf16x3"
run k_fp32_on2 "V3D_PRESPLIT=1" ""
run k_bf16_on2 "V3D_PRESPLIT=1" "--precision bf16x3"
run w_fp32_on "V3D_PRESPLIT=1" "--workload waymo"
run w_bf16_on "V3D_PRESPLIT=1" "--workload waymo --precision bf16x3"
run w_bf16_off "V3D_PRESPLIT=0" "--workload waymo --precision bf16x3"
