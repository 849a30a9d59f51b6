# one GPU pass: whole GPU suite, in-frame sequence timeline, pipelined bench (short + long windows)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r4c}
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/${T}_tests.txt
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python bench.py --pipeline 1 --steps 300 --warmup 20 --no-cpu-baseline --no-roofline --no-h2d --windows 1 > gpurun_out/${T}_seq_bench.json 2> /tmp/prof_seq.err
f=$(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1)
python tools/trace_sequence.py $f 200 > gpurun_out/${T}_trace_sequence.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench300.json 2>> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_tests.txt; head -60 gpurun_out/${T}_trace_sequence.txt; cut -c1-300 gpurun_out/${T}_bench.json; echo; cut -c1-300 gpurun_out/${T}_bench300.json; tail -3 gpurun_out/${T}_bench.err
