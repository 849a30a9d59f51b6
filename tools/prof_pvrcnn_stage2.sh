# On the GPU box (gpurun -- bash tools/prof_pvrcnn_stage2.sh): rocprofv3 kernel statistics of one bench / microbenchmark command, top kernels printed;
# the csv lands in gpurun_out/.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pvp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pvp -- python $GRAFT_REPO_ROOT/bench.py --mode pvrcnn --steps 64 --warmup 8 --no-cpu-baseline --no-roofline > /tmp/pvp.json 2>/tmp/pvp.err
cd $GRAFT_REPO_ROOT
tail -1 /tmp/pvp.json | cut -c1-300
cp $(find /tmp/pvp -name "*kernel_stats.csv" | head -1) gpurun_out/pv_stage2_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/pv_stage2_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print(len(rows), tot/1e6)
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.2f}")
PY
