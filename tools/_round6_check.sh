cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_plan.py tests/test_gpu_brick.py tests/test_gpu_second.py tests/test_gpu_configs.py tests/test_gpu_dense_conv.py tests/test_gpu_dropin.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
