cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmb
MB_VARIANTS=0 MB_SEQUENCE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmb -- python tools/mb_sparse_layers.py kitti > /tmp/pmb.log 2>&1
grep "64-> 64 K=27" /tmp/pmb.log | head -3
f=$(find /tmp/pmb -name "*kernel_stats.csv" | head -1)
python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "spconv_fwd_rows" in r["Name"]:
        print("%-70s calls %6s avg %6.2f us min %6.2f max %6.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
