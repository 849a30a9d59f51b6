#!/bin/bash
# Round 6: matrix-pipe counters of the kernels that SHIP (review item 4): SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVES,
# SQ_INSTS_VALU_MFMA_MOPS_F16 / _BF16, GRBM_GUI_ACTIVE per launch of the dominant sparse and dense kernels in the KITTI frame, the
# Waymo-range frame and the train step -- one counter per rocprofv3 pass (--pmc with --kernel-trace only, as the guide prescribes),
# summarised into gpurun_out/pmc_mfma.txt and gpurun_out/pmc_mfma.json (copy to profiles/r06_pmc_mfma.*).
# mfma_busy_frac of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x SIMDs that could be busy): reported two ways --
# against all 1 024 SIMDs for the launch's duration, and per busy SIMD-cycle (SQ_BUSY_CYCLES based).
# usage: bash tools/pmc_mfma.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp PYTHONUNBUFFERED=1; cd "$R"
COUNTERS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES FETCH_SIZE WRITE_SIZE"
run() {  # name, bench args...
  name=$1; shift
  for c in $COUNTERS; do
    rm -rf "/tmp/pmcm_${name}_$c"
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "/tmp/pmcm_${name}_$c" -o p -- python bench.py "$@" > "/tmp/pmcm_${name}_$c.log" 2>&1
    f=$(find "/tmp/pmcm_${name}_$c" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$R/gpurun_out/pmcm_${name}_$c.csv" || echo "no counters for $name $c: $(tail -2 /tmp/pmcm_${name}_$c.log)"
  done
}
COMMON="--no-cpu-baseline --no-fast-mode --no-h2d --no-extra --no-roofline --windows 1 --single-frames 10"
run kitti --steps 6 --warmup 3 --pipeline 1 $COMMON
run waymo --workload waymo --steps 4 --warmup 2 --pipeline 1 $COMMON
run train --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-extra
python tools/pmc_mfma.py "$R/gpurun_out" | tee "$R/gpurun_out/pmc_mfma.txt"
