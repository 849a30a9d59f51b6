# round-4 GPU pass H: plan / dense / proposal tests after the densify fusion; proposal chunk-count variants (one-frame kernel stats)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_plan.py tests/test_gpu_second.py tests/test_gpu_proposal.py tests/test_gpu_dropin.py tests/test_gpu_configs.py tests/test_gpu_dense_conv.py tests/test_gpu_conv3d_parity.py -x -q -s -m gpu 2>&1 | grep -E "passed|failed|Error|error|worst strict|layer  [0-9]|layer 1[0-3]" | tail -22 > gpurun_out/r4h_tests.txt
O=gpurun_out/r4h_variants.txt; : > $O
for lib in "" vision3d_amd/lib/libvision3d_hip_c20.so vision3d_amd/lib/libvision3d_hip_c12.so; do
  echo "== lib=${lib:-default}" >> $O
  rm -rf /tmp/prof_v
  V3D_HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_v -- python bench.py --pipeline 1 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-h2d --windows 1 > /tmp/v.json 2> /tmp/v.err
  f=$(find /tmp/prof_v -name "*kernel_trace.csv" | head -1)
  python tools/trace_sequence.py $f 100 | grep -E "frames of|prop_|rows<64, 64>|densify|nms_" >> $O
  python -c "
import json; d=json.loads(open('/tmp/v.json').read().strip().splitlines()[-1]); print('single_frame', d['single_frame'])" >> $O
done
cat gpurun_out/r4h_tests.txt; cat $O
