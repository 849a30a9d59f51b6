# Round-5 closing measurements on one MI355X box.  Order matters: the one-frame-at-a-time kernel statistics come FIRST and are
# copied into profiles/ on the box, because bench.py attaches them to its line as roofline.in_frame (the committed copies do the
# same for the driver's run).  Output: gpurun_out/closing5/ -> profiles/r05_<tag>_* (copy by hand, see profiles/README.md).
#   usage: bash tools/closing_r05.sh [tag] [quick]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; T=${1:-a}; O=gpurun_out/closing5; mkdir -p $O
stats() {  # stats <name> <bench args...>: rocprofv3 kernel statistics of one bench command -> $O/<name>_kernel_stats.csv
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python bench.py "$@" > $O/${name}_prof.json 2> $O/${name}_prof.err
  find /tmp/prof_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp $f /tmp/${name}_kernel_trace.csv
  rm -rf /tmp/prof_$name
}
# 1. one frame at a time: per-kernel statistics + the in-frame sequence timeline (KITTI, Waymo range)
stats one_frame --pipeline 1 --steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-h2d --no-fast-mode --windows 1
python tools/trace_sequence.py /tmp/one_frame_kernel_trace.csv 200 > $O/trace_sequence.txt 2>&1
cp $O/one_frame_kernel_stats.csv profiles/in_frame_kernel_stats.csv
stats waymo_one_frame --workload waymo --pipeline 1 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-h2d --no-fast-mode --windows 1
python tools/trace_sequence.py /tmp/waymo_one_frame_kernel_trace.csv 40 > $O/waymo_trace_sequence.txt 2>&1
cp $O/waymo_one_frame_kernel_stats.csv profiles/in_frame_kernel_stats_waymo.csv
# 2. the bench lines (the first two exactly as the driver runs them / as bench.py defaults)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
stats pipelined --steps 200 --warmup 20 --no-cpu-baseline --no-h2d --no-fast-mode
python tools/trace_overlap.py /tmp/pipelined_kernel_trace.csv > $O/trace_overlap.txt 2>&1
timeout 900 python bench.py --workload waymo --cpu-frames 1 > $O/waymo.json 2> $O/waymo.err
stats waymo --workload waymo --steps 40 --warmup 10 --no-cpu-baseline --no-h2d --no-fast-mode
timeout 900 python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err
stats train --mode train --steps 10 --warmup 3 --no-cpu-baseline
timeout 900 python bench.py --mode pvrcnn --steps 20 --warmup 5 > $O/pvrcnn.json 2> $O/pvrcnn.err
timeout 900 python bench.py --mode pvrcnn --end-to-end --steps 20 --warmup 5 > $O/pvrcnn_e2e.json 2> $O/pvrcnn_e2e.err
timeout 900 python bench.py --mode plumbing --steps 300 --warmup 30 > $O/plumbing.json 2> $O/plumbing.err
timeout 900 python bench.py --precision bf16x3 --no-cpu-baseline > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
if [ "${2:-}" != quick ]; then
  stats pvrcnn --mode pvrcnn --steps 10 --warmup 3 --pipeline 1 --no-cpu-baseline
  stats pvrcnn_e2e --mode pvrcnn --end-to-end --steps 10 --warmup 3 --no-cpu-baseline
  # 3. HBM traffic of the dominant kernels (two --pmc passes each, counters with --kernel-trace only)
  timeout 1500 bash tools/pmc_traffic.sh kitti r05 > $O/pmc_traffic_run.txt 2>&1; cp gpurun_out/pmc_traffic.txt $O/pmc_traffic.txt; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json
  timeout 1500 bash tools/pmc_traffic.sh waymo r05 > $O/pmc_traffic_waymo_run.txt 2>&1; cp gpurun_out/pmc_traffic_waymo.txt $O/pmc_traffic_waymo.txt; cp gpurun_out/pmc_traffic_waymo.json $O/pmc_traffic_waymo.json
  timeout 600 python tools/mb_prec_ab.py > $O/mb_prec_ab.txt 2>&1
fi
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE /tmp/*_kernel_trace.csv
for f in bench_driver_form bench waymo train pvrcnn pvrcnn_e2e; do echo "== $f"; cut -c1-330 $O/$f.json; tail -2 $O/$f.err; done
head -50 $O/trace_sequence.txt
