# round-4 GPU pass M: sanity of the secondary bench modes (batch 8, native / eager paths, waymo batch 2)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4m_modes.txt; : > $O
run() { echo "== $*" >> $O; python bench.py "$@" --no-cpu-baseline --no-roofline --no-h2d 2>> $O | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else {}
print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','single_frame_ms','n_proposals')}, (d.get('config') or {}).get('pipeline_tuning'))" >> $O; }
run --batch 8 --steps 50 --warmup 5
run --path native --steps 50 --warmup 5
run --path eager --steps 50 --warmup 5
run --pipeline 2 --steps 50 --warmup 5
run --workload waymo --batch 2 --steps 20 --warmup 3
grep -v "amdgpu.ids" $O
