cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/dt; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dense_conv.py tests/test_gpu_dense_train.py tests/test_gpu_sparse_conv.py -q -m gpu -x 2>&1 | tail -8 > $O/tests.txt; cat $O/tests.txt
timeout 900 python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-300 $O/train.json; tail -2 $O/train.err
