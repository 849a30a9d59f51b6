cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_sparse_conv.py -x -q -m gpu -k tiles_per_workgroup 2>&1 | grep -E "Error|error|assert|Mismatch|rows_hint|n=" | head -20
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python bench.py --pipeline 1 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-h2d --windows 1 > /dev/null 2> /tmp/prof_seq.err
python tools/trace_sequence.py $(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1) 100 | grep -E "frames of|ring"
