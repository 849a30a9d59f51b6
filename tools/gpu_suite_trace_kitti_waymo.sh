# one GPU pass: whole suite + KITTI / Waymo one-frame sequence timelines + bench lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r4d}
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/${T}_tests.txt
for wl in kitti waymo; do
  rm -rf /tmp/prof_seq
  extra="--steps 300 --warmup 20"; [ $wl = waymo ] && extra="--workload waymo --steps 60 --warmup 10"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python bench.py --pipeline 1 $extra --no-cpu-baseline --no-roofline --no-h2d --windows 1 > gpurun_out/${T}_${wl}_seq_bench.json 2> /tmp/prof_seq.err
  f=$(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1)
  python tools/trace_sequence.py $f 100 > gpurun_out/${T}_${wl}_trace_sequence.txt 2>&1
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --workload waymo --no-cpu-baseline --no-roofline > gpurun_out/${T}_waymo.json 2>> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_tests.txt; head -44 gpurun_out/${T}_kitti_trace_sequence.txt; head -30 gpurun_out/${T}_waymo_trace_sequence.txt; cut -c1-200 gpurun_out/${T}_bench.json; echo; cut -c1-200 gpurun_out/${T}_waymo.json; tail -3 gpurun_out/${T}_bench.err
