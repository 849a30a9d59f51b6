# Round-5 closing, part c: the bench lines as the tree stands at the end of the round (the kernel statistics / PMC passes of part b,
# tools/closing_r05.sh, cover kernels this part did not change).  Output: gpurun_out/closing5c/ -> profiles/r05_c_*.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/closing5c; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
timeout 900 python bench.py --workload waymo --cpu-frames 1 > $O/waymo_bench.json 2> $O/waymo.err
timeout 900 python bench.py --mode train --steps 20 --warmup 5 > $O/train_bench.json 2> $O/train.err
timeout 900 python bench.py --mode pvrcnn --steps 20 --warmup 5 > $O/pvrcnn_bench.json 2> $O/pvrcnn.err
timeout 900 python bench.py --mode pvrcnn --end-to-end --steps 30 --warmup 5 > $O/pvrcnn_e2e_bench.json 2> $O/pvrcnn_e2e.err
timeout 900 python bench.py --mode plumbing --steps 300 --warmup 30 > $O/plumbing_bench.json 2> $O/plumbing.err
rm -rf /tmp/prof_w; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_w -- python bench.py --workload waymo --steps 60 --warmup 10 --pipeline 1 --no-cpu-baseline --no-roofline --no-h2d --no-fast-mode --windows 1 > /dev/null 2>&1
python tools/trace_sequence.py $(find /tmp/prof_w -name "*kernel_trace.csv" | head -1) 40 > $O/waymo_trace_sequence.txt 2>&1; rm -rf /tmp/prof_w
for f in bench_driver_form waymo_bench train_bench pvrcnn_bench pvrcnn_e2e_bench plumbing_bench; do echo "== $f"; cut -c1-200 $O/$f.json; done
head -3 $O/waymo_trace_sequence.txt
