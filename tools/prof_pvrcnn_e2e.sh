# On the GPU box (gpurun -- bash tools/prof_pvrcnn_e2e.sh): rocprofv3 kernel statistics of one bench / microbenchmark command, top kernels printed;
# the csv lands in gpurun_out/.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/e2e; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e -- python $GRAFT_REPO_ROOT/bench.py --mode pvrcnn --end-to-end --steps 64 --warmup 8 --no-cpu-baseline --no-roofline > /tmp/e2e.json 2>/tmp/e2e.err
cd $GRAFT_REPO_ROOT
cp $(find /tmp/e2e -name "*kernel_stats.csv" | head -1) gpurun_out/pv_e2e_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/pv_e2e_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
calls=sum(int(r["Calls"]) for r in rows)
print(len(rows), tot/1e6, calls)
rows.sort(key=lambda r:-int(r["Calls"]))
for r in rows[:45]:
    print(f"{r['Name'][:120]:120s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f}")
PY
