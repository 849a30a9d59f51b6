# Round-5 closing, part c2: Waymo-range kernel statistics of the final tree (the submanifold-table change touched that frame only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/closing5c; mkdir -p $O
stats() {
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python bench.py "$@" > $O/${name}_prof.json 2> $O/${name}_prof.err
  find /tmp/prof_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
  rm -rf /tmp/prof_$name
}
stats waymo_one_frame --workload waymo --pipeline 1 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-h2d --no-fast-mode --windows 1
stats waymo --workload waymo --steps 40 --warmup 10 --no-cpu-baseline --no-h2d --no-fast-mode
head -8 $O/waymo_one_frame_kernel_stats.csv | cut -c1-60,150-260
