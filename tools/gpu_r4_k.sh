# round-4 GPU pass K: pipeline depth (V3D_BENCH_MAX_PIPELINE 4 / 6 / 8), same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4k_depth.txt; : > $O
for d in 4 6 8 4 6; do
  V3D_BENCH_MAX_PIPELINE=$d python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('max_depth $d ->', round(d['value'],1), 'p10', round(d['value_p10'],1), 'p90', round(d['value_p90'],1), d['config']['pipeline_tuning'])" >> $O
done
for d in 4 6; do
  V3D_BENCH_MAX_PIPELINE=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20-step windows, max_depth $d ->', round(d['value'],1), 'p10', round(d['value_p10'],1), 'p90', round(d['value_p90'],1), d['config']['pipeline_tuning'])" >> $O
done
cat $O
