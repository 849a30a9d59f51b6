cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15
( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | tail -3; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "single", d["single_frame_ms"], "roofline frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
for k,v in d["extra"].items(): print(k, json.dumps({a:b for a,b in v.items() if a not in ("what","workload")}))
PY
