# Round 5: fused augmentation (csrc/augment.hip) parity + timing; rulebook KPT switch re-check (KITTI must be back, Waymo kept).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/aug; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_augmentation.py tests/test_gpu_iou_nms.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
timeout 300 python tools/mb_augmentation.py > $O/mb_augmentation.txt 2>&1; cat $O/mb_augmentation.txt
rm -rf /tmp/prof_aug; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_aug -- python tools/mb_augmentation.py > /dev/null 2>&1
find /tmp/prof_aug -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/aug_kernel_stats.csv; head -12 $O/aug_kernel_stats.csv | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_sparse_conv.py tests/test_gpu_plan.py tests/test_gpu_second.py -x -q -m gpu 2>&1 | tail -3 > $O/tests_rb.txt; cat $O/tests_rb.txt
timeout 600 python bench.py --workload waymo --no-cpu-baseline > $O/waymo.json 2> $O/waymo.err
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/waymo.json $O/bench.json
