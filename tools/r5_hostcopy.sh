#!/bin/bash
mkdir -p gpurun_out/r5m
export PYTHONPATH=/root/repo:$PYTHONPATH
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_proposal.py -x -q -m gpu 2>&1 | tail -6
for v in 1 0 1 0; do
  V3D_GRAPH_HOST_COPY=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-fast-mode --no-roofline --windows 15 > gpurun_out/r5m/b$v.json 2> gpurun_out/r5m/b$v.err
  python - $v <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5m/b{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("host copy in graph =", sys.argv[1], "value %.0f (p10 %.0f p90 %.0f) single %.4f" % (d["value"], d["value_p10"], d["value_p90"], d["single_frame_ms"]))
PY
done
