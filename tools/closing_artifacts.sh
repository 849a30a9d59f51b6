# Round-closing measurements on one MI355X box (usage, on the GPU box: ROUND=r06 bash tools/closing_artifacts.sh): the bench lines of
# every BASELINE.json configuration with their rocprofv3 kernel statistics, the one-frame kernel sequences, the pipeline overlap
# trace and the HBM-traffic counter passes of the dominant sparse kernels.  Output: gpurun_out/closing_$ROUND/ ; afterwards, in the
# build container:  ROUND=r06 bash tools/closing_artifacts.sh copy   puts the summaries under profiles/ (tracked).
ROUND=${ROUND:-r06}; SET=${SET:-c}  # SET: letter of the closing set under profiles/ (r06_c_*: mid-round, r06_d_*: final tree)
if [ "${1:-}" = "copy" ]; then
  O=gpurun_out/closing_$ROUND
  for f in bench bench_driver_form waymo train train_fp32_script pvrcnn pvrcnn_e2e plumbing; do [ -s $O/$f.json ] && cp $O/$f.json profiles/${ROUND}_${SET}_${f}.json; done
  for f in kernel_stats one_frame_at_a_time_kernel_stats waymo_kernel_stats waymo_one_frame_at_a_time_kernel_stats train_kernel_stats pvrcnn_stage2_kernel_stats pvrcnn_e2e_kernel_stats; do
    [ -s $O/$f.csv ] && cp $O/$f.csv profiles/${ROUND}_${SET}_${f}.csv
  done
  for f in trace_overlap trace_sequence waymo_trace_sequence; do [ -s $O/$f.txt ] && cp $O/$f.txt profiles/${ROUND}_${SET}_${f}.txt; done
  [ -s $O/one_frame_at_a_time_kernel_stats.csv ] && cp $O/one_frame_at_a_time_kernel_stats.csv profiles/in_frame_kernel_stats.csv
  [ -s $O/waymo_one_frame_at_a_time_kernel_stats.csv ] && cp $O/waymo_one_frame_at_a_time_kernel_stats.csv profiles/in_frame_kernel_stats_waymo.csv
  [ -s gpurun_out/pmc_traffic.txt ] && cp gpurun_out/pmc_traffic.txt profiles/${ROUND}_pmc_traffic.txt && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
  [ -s gpurun_out/pmc_traffic_waymo.txt ] && cp gpurun_out/pmc_traffic_waymo.txt profiles/${ROUND}_pmc_traffic_waymo.txt && cp gpurun_out/pmc_traffic_waymo.json profiles/pmc_traffic_waymo.json
  ls profiles | grep "^${ROUND}_" | tr '\n' ' '
  exit 0
fi
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/closing_$ROUND; mkdir -p $O
stats() {  # name, bench args...: rocprofv3 --kernel-trace --stats of one bench command -> $O/<name>.csv
  n=$1; shift
  rm -rf /tmp/prof_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -- python bench.py "$@" > /tmp/prof_$n.json 2> /tmp/prof_$n.err
  find /tmp/prof_$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/$n.csv
  rm -rf /tmp/prof_$n
}
seq() {  # name, bench args...: the one-frame kernel sequence timeline
  n=$1; shift
  rm -rf /tmp/prof_$n
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$n -- python bench.py "$@" > /dev/null 2> /tmp/prof_$n.err
  python tools/trace_sequence.py $(find /tmp/prof_$n -name "*kernel_trace.csv" | head -1) 100 > $O/$n.txt 2>&1
  rm -rf /tmp/prof_$n
}
Q="--no-cpu-baseline --no-extra"
python bench.py > $O/bench_driver_form.json 2> $O/bench_driver_form.err                      # exactly what the driver runs
python bench.py --steps 300 --warmup 30 --no-extra > $O/bench.json 2> $O/bench.err
stats kernel_stats --steps 200 --warmup 20 $Q --no-fast-mode --no-h2d
stats one_frame_at_a_time_kernel_stats --pipeline 1 --steps 200 --warmup 20 $Q --no-fast-mode --no-h2d --no-roofline
seq trace_sequence --pipeline 1 --steps 300 --warmup 20 $Q --no-roofline --no-h2d --no-fast-mode --windows 1
rm -rf /tmp/prof_o; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_o -- python bench.py --steps 200 --warmup 20 $Q --no-roofline --no-fast-mode --no-h2d > /dev/null 2> /tmp/prof_o.err
python tools/trace_overlap.py $(find /tmp/prof_o -name "*kernel_trace.csv" | head -1) > $O/trace_overlap.txt 2>&1; rm -rf /tmp/prof_o
python bench.py --workload waymo --no-extra > $O/waymo.json 2> $O/waymo.err
stats waymo_kernel_stats --workload waymo --steps 40 --warmup 10 $Q --no-fast-mode --no-h2d
stats waymo_one_frame_at_a_time_kernel_stats --workload waymo --pipeline 1 --steps 40 --warmup 10 $Q --no-fast-mode --no-h2d --no-roofline
seq waymo_trace_sequence --workload waymo --pipeline 1 --steps 60 --warmup 10 $Q --no-roofline --no-h2d --no-fast-mode --windows 1
python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err
stats train_kernel_stats --mode train --steps 10 --warmup 3 --no-cpu-baseline
python bench.py --mode pvrcnn --steps 20 --warmup 5 > $O/pvrcnn.json 2> $O/pvrcnn.err
stats pvrcnn_stage2_kernel_stats --mode pvrcnn --steps 10 --warmup 3 --no-cpu-baseline
python bench.py --mode pvrcnn --end-to-end --steps 20 --warmup 5 > $O/pvrcnn_e2e.json 2> $O/pvrcnn_e2e.err
stats pvrcnn_e2e_kernel_stats --mode pvrcnn --end-to-end --steps 10 --warmup 3 --no-cpu-baseline
python bench.py --mode plumbing > $O/plumbing.json 2> $O/plumbing.err
bash tools/pmc_traffic.sh kitti $ROUND > $O/pmc_traffic.log 2>&1
bash tools/pmc_traffic.sh waymo $ROUND > $O/pmc_traffic_waymo.log 2>&1
for f in bench_driver_form bench waymo train pvrcnn pvrcnn_e2e plumbing; do cut -c1-260 $O/$f.json; echo; done
head -45 $O/trace_sequence.txt
