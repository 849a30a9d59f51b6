# round-4 GPU pass P: throughput mode of the pipelined slots: tests + bench lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4p.txt; : > $O
python -m pytest tests/test_gpu_proposal.py tests/test_gpu_plan.py tests/test_gpu_sparse_conv.py -x -q -m gpu 2>&1 | tail -2 >> $O
for rep in 1 2; do for st in 20 300; do
python bench.py --steps $st --warmup 5 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $st ->', round(d['value'],1), 'p10', round(d['value_p10'],1), 'p90', round(d['value_p90'],1), 'single', round(d['single_frame_ms'],4), d['config']['pipeline_tuning'])" >> $O
done; done
python bench.py --workload waymo --steps 300 --warmup 5 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waymo ->', round(d['value'],1), 'single', round(d['single_frame_ms'],4), d['config']['pipeline_tuning'])" >> $O
cat $O
