# round-4 GPU pass Q: persistent grid of the dense tile kernel (256 / 192 / 128 workgroups) vs pipelined throughput, same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r4q_dense_grid.txt; : > $O
for rep in 1 2; do for lib in "" vision3d_amd/lib/libvision3d_hip_g192.so vision3d_amd/lib/libvision3d_hip_g128.so; do
  V3D_HIP_LIB=$lib python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-default} ->', round(d['value'],1), 'p10', round(d['value_p10'],1), 'p90', round(d['value_p90'],1), 'single', round(d['single_frame_ms'],4), d['config']['pipeline_tuning'])" >> $O
done; done
cat $O
