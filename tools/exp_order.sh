# Round-6 experiment 0: what spatial locality of the ROW ORDER alone does to the existing kernels (no new code): the Waymo-range
# sweep with its returns shuffled (default) / Morton-sorted on the host, so that the voxelizer's first-touch row order is a Morton
# order.  Output: gpurun_out/order_*.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for o in shuffled morton; do
  python bench.py --workload waymo --order $o --no-cpu-baseline --no-fast-mode --no-h2d --windows 7 --steps 100 > gpurun_out/order_$o.json 2> gpurun_out/order_$o.err
  bash tools/prof_stats.sh order_${o}_1f --workload waymo --order $o --pipeline 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-roofline --windows 3 --steps 60 > gpurun_out/order_${o}_prof.txt 2>&1
done
python - <<'PY'
import json
for o in ("shuffled", "morton"):
    try:
        d = json.loads(open(f"gpurun_out/order_{o}.json").read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(o, "value", round(d["value"], 1), "single_frame_ms", d.get("single_frame_ms"), "kouter avg_us", r.get("avg_us"), [round(l["t_avg_us"], 1) for l in d.get("stages", {}).get("layers", [])])
    except Exception as e:
        print(o, "failed", e)
PY
for o in shuffled morton; do echo == $o; head -45 gpurun_out/order_${o}_prof.txt; done
