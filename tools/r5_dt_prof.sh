cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/dt; mkdir -p $O
rm -rf /tmp/prof_t; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -- python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-roofline > $O/train_prof.json 2> $O/train_prof.err
find /tmp/prof_t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_fp32_kernel_stats.csv; head -30 $O/train_fp32_kernel_stats.csv | cut -c1-70,160-260
