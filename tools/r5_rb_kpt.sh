# Round 5: submanifold tables with KPT offsets per thread (rulebook.hip rb_subm_entries).  Rulebook parity tests first, then the
# in-frame sequences (Waymo range, KITTI) and the bench lines.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/rbkpt; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_conv.py tests/test_gpu_plan.py tests/test_gpu_second.py tests/test_gpu_conv3d_parity.py -x -q -m gpu 2>&1 | tail -4 > $O/tests.txt
seq() {  # seq <name> <frames> <bench args>
  local name=$1 frames=$2; shift 2
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$name -- python bench.py "$@" --pipeline 1 --no-cpu-baseline --no-roofline --no-h2d --no-fast-mode --windows 1 > $O/${name}_prof.json 2> $O/${name}_prof.err
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  python tools/trace_sequence.py $f $frames > $O/${name}_sequence.txt 2>&1
  rm -rf /tmp/prof_$name
}
seq waymo 40 --workload waymo --steps 60 --warmup 10
seq kitti 200 --steps 300 --warmup 20
timeout 600 python bench.py --workload waymo --no-cpu-baseline > $O/waymo.json 2> $O/waymo.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/tests.txt; grep "rb_\|frames of" $O/waymo_sequence.txt $O/kitti_sequence.txt | cut -c1-140; cut -c1-300 $O/waymo.json $O/bench.json
