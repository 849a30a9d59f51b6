#!/bin/bash
# round-5 GPU lease: the f16s arithmetic -- parity tests of the touched paths, then timings of both arithmetics
mkdir -p gpurun_out/r5d
export PYTHONPATH=/root/repo:$PYTHONPATH
./tools/mb_f16split > gpurun_out/r5d/f16split.txt 2>&1; head -8 gpurun_out/r5d/f16split.txt
timeout 1200 python -m pytest tests/test_gpu_plan.py tests/test_gpu_dense_conv.py tests/test_gpu_conv3d_parity.py tests/test_gpu_sparse_conv.py tests/test_gpu_second.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r5d/pytest.txt
tail -15 gpurun_out/r5d/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5d/bench_fp32.json 2> gpurun_out/r5d/bench_fp32.err
tail -c 1500 gpurun_out/r5d/bench_fp32.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5d/bench_fp32.json").read().strip().splitlines()[-1])
    print("value", d["value"], "single_ms", d["single_frame_ms"], "fast", {k: d["fast_mode"].get(k) for k in ("value", "single_frame_ms", "error")})
    print("roofline", {k: d["roofline"][k] for k in ("kernel", "avg_us", "frac", "bound")})
    print("dense", {k: d["roofline_dense"][k] for k in ("avg_us", "frac")})
    print("stages", {k: v for k, v in d["stages"].items() if k not in ("layers", "backbone_note")})
    print([l["t_avg_us"] for l in d["stages"]["layers"]])
except Exception as e:
    print("bench parse failed", e)
PY
