#!/bin/bash
mkdir -p gpurun_out/r5l
export PYTHONPATH=/root/repo:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_sparse_conv.py -x -q -m gpu -k "presplit or f16s_scales" 2>&1 | tail -8
timeout 300 python tools/mb_prec_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5l/mb_prec_ab.txt | tail -16
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --windows 9 > gpurun_out/r5l/bench.json 2> gpurun_out/r5l/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5l/bench.json").read().strip().splitlines()[-1])
print("value %.0f single %.4f fast %s" % (d["value"], d["single_frame_ms"], {k: d["fast_mode"].get(k) for k in ("value", "single_frame_ms", "error")}))
print({k: d["roofline"][k] for k in ("kernel", "avg_us", "frac", "bound")}, d["stages"]["sparse_conv_us"])
print([l["t_avg_us"] for l in d["stages"]["layers"]])
PY
