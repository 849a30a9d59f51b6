"""Summarise a rocprofv3 kernel-trace CSV: per-kernel stats restricted to the steady-state window, GPU
busy time and gaps per step.  usage: trace_summary.py <kernel_trace.csv> <steps_in_window>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady state = last `steps` occurrences of the voxelizer insert kernel mark step starts
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("vox_insert_kernel")]
start = marks[-steps] if len(marks) >= steps else 0
win = rows[start:]
t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    a = agg[r["Kernel_Name"][:90]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"window: {steps} steps, wall {(t1 - t0) / steps / 1e3:.1f} us/step, GPU busy {busy / steps / 1e3:.1f} us/step, "
      f"{len(win) / steps:.1f} kernels/step")
print(f"{'kernel':92s} {'n/step':>7s} {'avg_us':>8s} {'us/step':>8s}")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k:92s} {n / steps:7.1f} {t / n / 1e3:8.2f} {t / steps / 1e3:8.1f}")
