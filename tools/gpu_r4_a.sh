# round-4 GPU pass A: launcher tests + changed-path tests, bench sanity, in-frame sequence timeline (one frame at a time)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_launch.py tests/test_gpu_proposal_loss.py tests/test_gpu_dense_train.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4a_tests.txt
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python bench.py --pipeline 1 --steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-h2d --windows 1 > gpurun_out/r4a_seq_bench.json 2> /tmp/prof_seq.err
f=$(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1)
python tools/trace_sequence.py $f 200 > gpurun_out/r4a_trace_sequence.txt 2>&1
tail -3 /tmp/prof_seq.err >> gpurun_out/r4a_trace_sequence.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
cat gpurun_out/r4a_tests.txt; head -70 gpurun_out/r4a_trace_sequence.txt; cut -c1-600 gpurun_out/r4a_bench.json
