cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/dt; mkdir -p $O
timeout 900 python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-900 $O/train.json; tail -3 $O/train.err
timeout 900 python -m pytest tests/test_gpu_dense_train.py tests/test_gpu_proposal_loss.py tests/test_gpu_bench_launch.py tests/test_gpu_dropin.py -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
