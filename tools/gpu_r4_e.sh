# round-4 GPU pass E: 3-term vs 4-term split product of the sparse kernels, whole frame A/B on one box + strict parity figures
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r4e_terms.txt; : > $O
for rep in 1 2; do
  for lib in "" vision3d_amd/lib/libvision3d_hip_t4.so; do
    echo "== lib=${lib:-default(3 terms)} rep=$rep" >> $O
    V3D_HIP_LIB=$lib python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline --no-h2d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined', round(d['value'],1), 'p10', round(d['value_p10'],1), 'p90', round(d['value_p90'],1), 'single_ms', round(d['single_frame_ms'],4), d['single_frame']['p10_ms'], d['single_frame']['p90_ms'])" >> $O
  done
done
for lib in "" vision3d_amd/lib/libvision3d_hip_t4.so; do
  echo "== strict parity, lib=${lib:-default(3 terms)}" >> $O
  V3D_HIP_LIB=$lib python -m pytest tests/test_gpu_conv3d_parity.py -x -q -s -m gpu 2>&1 | grep -E "layer|passed|failed|Error" >> $O
  V3D_HIP_LIB=$lib python __graft_entry__.py smoke 2>&1 | tail -1 >> $O
done
cat $O
