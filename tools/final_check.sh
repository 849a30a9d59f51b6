cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cat gpurun_out/final_tests.txt; tail -2 gpurun_out/final_smoke.txt; cut -c1-400 gpurun_out/final_bench.json
