cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sparse_conv.py tests/test_gpu_plan.py tests/test_gpu_configs.py tests/test_gpu_conv3d_parity.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -8
MB_ORDERS=morton,shuffled timeout 300 python tools/mb_brick.py waymo 2>&1 | grep -E "^==|->|rror" | sed -e "s/| union.*| plan/| plan/"
timeout 600 python bench.py --workload waymo --no-cpu-baseline --no-fast-mode --no-h2d --windows 7 --steps 100 > gpurun_out/waymo_epi.json 2> gpurun_out/waymo_epi.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/waymo_epi.json").read().strip().splitlines()[-1])
print("waymo value", round(d["value"], 1), "single_frame_ms", d.get("single_frame_ms"), [round(l["t_avg_us"], 1) for l in d.get("stages", {}).get("layers", [])])
PY
timeout 600 python bench.py --no-cpu-baseline --no-fast-mode --no-h2d --windows 7 --steps 100 > gpurun_out/kitti_epi.json 2> gpurun_out/kitti_epi.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/kitti_epi.json").read().strip().splitlines()[-1])
print("kitti value", round(d["value"], 1), "single_frame_ms", d.get("single_frame_ms"), [round(l["t_avg_us"], 1) for l in d.get("stages", {}).get("layers", [])])
PY
