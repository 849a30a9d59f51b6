cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sparse_conv.py tests/test_gpu_plan.py tests/test_gpu_conv3d_parity.py tests/test_gpu_configs.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-fast-mode --no-h2d --windows 9 --steps 200 > gpurun_out/b.json 2> gpurun_out/b.err; tail -2 gpurun_out/b.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "single", d["single_frame_ms"], [round(l["t_avg_us"],1) for l in d["stages"]["layers"]])
for k,v in d["extra"].items(): print("  extra", k, v.get("value"), v.get("seconds"), v.get("error"))
PY
python bench.py --workload waymo --no-cpu-baseline --no-fast-mode --no-h2d --no-extra --windows 7 --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waymo value', round(d['value'],1), 'single', d['single_frame_ms'], [round(l['t_avg_us'],1) for l in d['stages']['layers']])"
