#!/bin/bash
# HBM traffic per launch of the two dominant kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters
# only with --kernel-trace, as MI355X_MICROARCH.md prescribes), summarised into gpurun_out/pmc_traffic.txt and .json.
# Copy both into profiles/ (r01_*_pmc_traffic.txt, pmc_traffic.json) after the run.
# usage: bash tools/pmc_traffic.sh [kitti|waymo] [tag]
set -u
WL=${1:-kitti}
TAG=${2:-r03}
EXTRA=""
[ "$WL" = "waymo" ] && EXTRA="--workload waymo"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd "$R"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$R/gpurun_out/pmc_$c"
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_$c" -o p -- \
    python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fast-mode --no-h2d --no-extra --windows 1 $EXTRA > "$R/gpurun_out/pmc_$c.log" 2>&1
  echo "pmc $c rc=$?"
done
SUF=""
[ "$WL" = "waymo" ] && SUF="_waymo"
python tools/pmc_traffic.py "$R/gpurun_out" "$WL" "$TAG" | tee "$R/gpurun_out/pmc_traffic$SUF.txt"
