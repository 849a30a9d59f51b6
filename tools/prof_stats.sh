# usage (on the GPU box): bash tools/prof_stats.sh <name> <bench args...>: rocprofv3 kernel statistics of one bench.py command,
# top kernels printed and the csv kept in gpurun_out/<name>_kernel_stats.csv
name=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python bench.py "$@" > gpurun_out/${name}_bench.json 2> /tmp/prof_$name.err
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { tail -20 /tmp/prof_$name.err; exit 1; }
cp $f gpurun_out/${name}_kernel_stats.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels: %d names, total %.2f ms" % (len(rows), tot / 1e6))
for r in rows[:34]:
    print("%-100s %6s calls %9.1f us avg %8.2f ms %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
cut -c1-300 gpurun_out/${name}_bench.json
