# Round 5: PV-RCNN end to end with the keypoint sampling on a side stream + grid-wide scale entries; augmentation edge test re-run.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/pv; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pointops.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
timeout 600 python bench.py --mode pvrcnn --end-to-end --steps 30 --warmup 5 > $O/pvrcnn_e2e.json 2> $O/pvrcnn_e2e.err
timeout 600 python bench.py --mode pvrcnn --steps 20 --warmup 5 > $O/pvrcnn.json 2> $O/pvrcnn.err
cut -c1-260 $O/pvrcnn_e2e.json $O/pvrcnn.json; tail -3 $O/pvrcnn_e2e.err
rm -rf /tmp/prof_pv; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pv -- python bench.py --mode pvrcnn --end-to-end --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
find /tmp/prof_pv -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/pvrcnn_e2e_kernel_stats.csv; head -8 $O/pvrcnn_e2e_kernel_stats.csv | cut -c1-60,200-300
