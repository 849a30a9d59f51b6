# round-4 GPU pass B: parity of the changed kernels, in-frame sequence timeline, pipelined bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_sparse_conv.py tests/test_gpu_proposal.py tests/test_gpu_iou_nms.py tests/test_gpu_second.py tests/test_gpu_plan.py tests/test_gpu_conv3d_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4b_tests.txt
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python bench.py --pipeline 1 --steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-h2d --windows 1 > gpurun_out/r4b_seq_bench.json 2> /tmp/prof_seq.err
f=$(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1)
python tools/trace_sequence.py $f 200 > gpurun_out/r4b_trace_sequence.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline > gpurun_out/r4b_bench300.json 2>> gpurun_out/r4b_bench.err
cat gpurun_out/r4b_tests.txt; head -60 gpurun_out/r4b_trace_sequence.txt; cut -c1-400 gpurun_out/r4b_bench.json; echo; cut -c1-400 gpurun_out/r4b_bench300.json; tail -3 gpurun_out/r4b_bench.err
