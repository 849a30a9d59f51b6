#!/bin/bash
# round-5 GPU lease: the f16s arithmetic -- parity tests of the touched paths, then timings of both arithmetics
mkdir -p gpurun_out/r5c
export PYTHONPATH=/root/repo:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_dense_conv.py tests/test_gpu_plan.py tests/test_gpu_second.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -60 > gpurun_out/r5c/pytest.txt
tail -30 gpurun_out/r5c/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5c/bench_fp32.json 2> gpurun_out/r5c/bench_fp32.err
tail -c 3000 gpurun_out/r5c/bench_fp32.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5c/bench_fp32.json").read().strip().splitlines()[-1])
    print("value", d["value"], "single_ms", d["single_frame_ms"], "fast", d.get("fast_mode"))
    print("roofline", {k: d["roofline"][k] for k in ("kernel", "avg_us", "frac", "bound")})
    print("dense", {k: d["roofline_dense"][k] for k in ("avg_us", "frac")})
    print("stages", {k: v for k, v in d["stages"].items() if k != "layers"})
except Exception as e:
    print("bench parse failed", e)
PY
