# FETCH_SIZE / WRITE_SIZE / LDS bank conflicts of the dense-train kernels in isolation (tools/mb_dense_train.py), one counter per pass.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python tools/mb_dense_train.py > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python - $f $c <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if k.startswith("dt_") or "dt_" in k:
        print("%-20s %-62s n=%3d mean=%.4g" % (sys.argv[2], k, len(v), sum(v) / len(v)))
PY
done
