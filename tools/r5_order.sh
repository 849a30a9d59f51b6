#!/bin/bash
mkdir -p gpurun_out/r5h
export PYTHONPATH=/root/repo:$PYTHONPATH
run() { # name, args
  timeout 400 python bench.py --no-cpu-baseline --no-fast-mode --no-h2d --windows 9 $2 > gpurun_out/r5h/$1.json 2> gpurun_out/r5h/$1.err
  python - "$1" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r5h/{n}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(n, "value %.0f single_ms %.4f | %s avg_us %.2f in hbm frac %.3f mfma_alg frac %.3f | sparse_us %.1f backbone %.1f" % (
        d["value"], d["single_frame_ms"], r["kernel"], r["avg_us"], r["hbm_view"]["frac"], r["mfma_view"]["frac_algorithmic"],
        d["stages"]["sparse_conv_us"], d["stages"]["backbone_us_with_voxelizer_and_rulebook_build"]))
    print("   ", [(l["n_out"], l["t_avg_us"]) for l in d["stages"]["layers"]])
except Exception as e:
    print(n, "failed", e); print(open(f"gpurun_out/r5h/{n}.err").read()[-1500:])
PY
}
run kitti_shuffled "--steps 20 --warmup 5"
run kitti_scan "--steps 20 --warmup 5 --order scan"
run waymo_shuffled "--workload waymo --steps 20 --warmup 5"
run waymo_scan "--workload waymo --steps 20 --warmup 5 --order scan"
run waymo_shuffled_bf16x3 "--workload waymo --steps 20 --warmup 5 --precision bf16x3"
