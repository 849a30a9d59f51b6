cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'single_ms',d.get('single_frame_ms'),'roofline frac',d['roofline']['frac'],'avg_us',d['roofline']['avg_us'])
for k,v in (d.get('extra') or {}).items(): print(k, round(v['value'],1), v.get('single_frame_ms'))
print('fast',d.get('fast_mode',{}).get('value'))
PY
