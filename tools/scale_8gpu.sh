#!/bin/bash
# Scaling runs on one 8-GPU MI355X node (the builder's boxes have ONE GPU: this script is what the driver / a maintainer runs
# where a node exists).  One process per GPU over RCCL (backend "nccl"), frames sharded by rank, no data-path collective for
# the forward lines; the train line adds the two-bucket gradient all-reduce (vision3d_amd/dist_util.py).  Every JSON line
# carries n_ranks_seen = an all-reduce of ones over the job: it must equal N.
#   usage: bash tools/scale_8gpu.sh [out_dir]      (env: NGPUS="1 2 4 8", STEPS, WARMUP, PORT)
set -u
OUT=${1:-gpurun_out/scale}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0   # the host driver only supports dmabuf IPC (RCCL across processes needs it)
NGPUS=${NGPUS:-"1 2 4 8"}; STEPS=${STEPS:-300}; WARMUP=${WARMUP:-30}; PORT=${PORT:-29511}
run() {  # run <n> <tag> <bench args...>
  local n=$1 tag=$2; shift 2
  # bench.py starts the N ranks itself when no launcher is around it (bench.launch_plan: torch.distributed.run, one process
  # per GPU); LAUNCHER=1 wraps it explicitly, the form the driver uses
  if [ "$n" = 1 ] || [ "${LAUNCHER:-0}" != 1 ]; then
    python bench.py --gpus "$n" "$@" > "$OUT/${tag}_n$n.json" 2> "$OUT/${tag}_n$n.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$n" "$@" > "$OUT/${tag}_n$n.json" 2> "$OUT/${tag}_n$n.err"
  fi
  python - "$OUT/${tag}_n$n.json" "$n" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ok = d.get("n_ranks_seen", d["n_gpus"]) == int(sys.argv[2]) == d["n_gpus"]
    print(f"{sys.argv[1]}: {d['value']:.1f} {d['unit']} on {d['n_gpus']} GPU(s), ranks seen {d.get('n_ranks_seen')}{'' if ok else '  <-- RANK COUNT MISMATCH'}")
except Exception as e:
    print(f"{sys.argv[1]}: no JSON line ({e})")
PY
}
for n in $NGPUS; do
  run "$n" forward --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline          # configs[1]: SECOND forward, KITTI cloud
  run "$n" waymo   --workload waymo --steps 100 --warmup 20 --no-cpu-baseline       # configs[4]: Waymo-range sweep
  run "$n" train   --mode train --steps 20 --warmup 5                               # configs[2]: train step bs = 8 / GPU + all-reduce
done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_n*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    tag = os.path.basename(f).rsplit("_n", 1)[0]
    rows.setdefault(tag, {})[d["n_gpus"]] = d["value"]
for tag, v in rows.items():
    base = v.get(1)
    print(tag, {n: (round(x, 1), None if not base else round(x / (n * base), 3)) for n, x in sorted(v.items())}, "(value, efficiency vs N x 1-GPU)")
PY
