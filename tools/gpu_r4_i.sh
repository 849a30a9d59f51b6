# round-4 GPU pass I: dense-head tests (fused 1x1 + head tail), whole-frame tests, KITTI sequence timeline, bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense_conv.py tests/test_gpu_second.py tests/test_gpu_proposal.py tests/test_gpu_dropin.py tests/test_gpu_pointops.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4i_tests.txt
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python bench.py --pipeline 1 --steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-h2d --windows 1 > gpurun_out/r4i_seq_bench.json 2> /tmp/prof_seq.err
python tools/trace_sequence.py $(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1) 100 > gpurun_out/r4i_trace_sequence.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline > gpurun_out/r4i_bench300.json 2>> gpurun_out/r4i_bench.err
cat gpurun_out/r4i_tests.txt; grep -E "frames of|conv|prop_|densify|rows<64" gpurun_out/r4i_trace_sequence.txt; cut -c1-200 gpurun_out/r4i_bench.json; echo; cut -c1-200 gpurun_out/r4i_bench300.json; tail -2 gpurun_out/r4i_bench.err
