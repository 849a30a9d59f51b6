# round-4 GPU pass L: pipeline tuned on windows of --steps frames: driver form (20) and default (300), repeated
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4l_tune.txt; : > $O
for rep in 1 2 3; do
  for st in 20 300; do
    python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $st ->', round(d['value'],1), 'p10', round(d['value_p10'],1), 'p90', round(d['value_p90'],1), d['config']['pipeline_tuning'])" >> $O
  done
done
python -m pytest tests/test_gpu_proposal.py -x -q -m gpu 2>&1 | tail -2 >> $O
cat $O
