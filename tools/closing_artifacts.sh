# Round-closing measurements on one MI355X box: the four bench lines of BASELINE.json with their rocprofv3 kernel statistics, the
# pipeline overlap trace and the microbenchmarks docs/rounds/design_rounds_1-4.md section 5b quotes.  Output: gpurun_out/closing/ (copied to profiles/r03_*).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/closing; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof_k.err
find $O/prof_k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $O/prof_o -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > /dev/null 2> $O/prof_o.err
python tools/trace_overlap.py $(find $O/prof_o -name "*kernel_trace.csv" | head -1) > $O/trace_overlap.txt 2>&1
python bench.py --workload waymo --no-cpu-baseline > $O/waymo.json 2> $O/waymo.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w -- python bench.py --workload waymo --steps 40 --warmup 10 --no-cpu-baseline > $O/waymo_prof.json 2> $O/prof_w.err
find $O/prof_w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/waymo_kernel_stats.csv
python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t -- python bench.py --mode train --steps 10 --warmup 3 > $O/train_prof.json 2> $O/prof_t.err
find $O/prof_t -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_kernel_stats.csv
python bench.py --mode pvrcnn --steps 20 --warmup 5 > $O/pvrcnn.json 2> $O/pvrcnn.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_p -- python bench.py --mode pvrcnn --steps 10 --warmup 3 > $O/pvrcnn_prof.json 2> $O/prof_p.err
find $O/prof_p -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/pvrcnn_kernel_stats.csv
# the full PV_RCNN.inference (stage 1 included): bench line + the kernels it runs
python bench.py --mode pvrcnn --end-to-end --steps 20 --warmup 5 > $O/pvrcnn_e2e.json 2> $O/pvrcnn_e2e.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -- python bench.py --mode pvrcnn --end-to-end --steps 10 --warmup 3 > $O/pvrcnn_e2e_prof.json 2> $O/prof_q.err
find $O/prof_q -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/pvrcnn_e2e_kernel_stats.csv
python tools/mb_dense_train.py > $O/mb_dense_train.txt 2>&1
bash tools/pmc_dense.sh > $O/pmc_dense_train.txt 2>&1
python tools/mb_bg_skip.py > $O/mb_bg_skip.txt 2>&1
python tools/mb_pipeline_host.py > $O/mb_pipeline_host.txt 2>&1
python tools/mb_sparse_layers.py kitti > $O/mb_sparse_layers.txt 2>&1
find $O -type d -name "prof_*" | xargs rm -rf
for f in bench waymo train pvrcnn pvrcnn_e2e; do cut -c1-260 $O/$f.json; done
