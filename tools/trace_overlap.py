"""Concurrency view of a rocprofv3 kernel-trace CSV of the pipelined bench: inside the window where the most queues are
active, how much of the wall time is covered by (a) any kernel, (b) a dense convolution, (c) a sparse convolution, and how
many kernels run at once.  usage: trace_overlap.py <kernel_trace.csv> [window_ms]"""
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
win_ns = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 5_000_000
# pick the window (stepping 1 ms) with the most distinct queues and then the most vox_insert launches (= frames)
t_min, t_max = rows[0][0], rows[-1][1]
best = None
t = t_min
idx = 0
while t + win_ns <= t_max:
    sel = [r for r in rows if r[0] >= t and r[1] <= t + win_ns]
    q = len({r[3] for r in sel})
    f = sum(1 for r in sel if r[2].startswith("vox_insert_kernel"))
    key = (q, f)
    if best is None or key > best[0]:
        best = (key, t, sel)
    t += 1_000_000
(qn, frames), t0, sel = best
print(f"window {win_ns/1e6:.1f} ms at +{(t0-t_min)/1e6:.1f} ms: {qn} queues, {frames} frames -> {win_ns/1e3/frames:.1f} us/frame")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
cats = {"any kernel": lambda n: True, "dense conv (conv2d_*)": lambda n: "conv2d_" in n, "sparse conv (spconv_*)": lambda n: "spconv_" in n,
        "dense or sparse conv": lambda n: "conv2d_" in n or "spconv_" in n,
        "rulebook/voxelizer/proposal/other": lambda n: not ("conv2d_" in n or "spconv_" in n)}
for name, pred in cats.items():
    iv = [(s, e) for s, e, n, q in sel if pred(n)]
    print(f"  {name:38s} covered {100*union(iv)/win_ns:5.1f} % of wall, sum of durations {sum(e-s for s,e in iv)/1e3/frames:7.1f} us/frame")
# concurrency histogram
ev = []
for s, e, n, q in sel: ev += [(s, 1), (e, -1)]
ev.sort(); cur = 0; last = t0; hist = collections.Counter()
for tt, d in ev:
    hist[cur] += tt - last; last = tt; cur += d
hist[cur] += t0 + win_ns - last
print("  kernels running at once: " + ", ".join(f"{k}: {100*v/win_ns:.1f} %" for k, v in sorted(hist.items())))
# dense-dense overlap: time with >= 2 dense kernels
ev = []
for s, e, n, q in sel:
    if "conv2d_bf16x3_large" in n: ev += [(s, 1), (e, -1)]
ev.sort(); cur = 0; last = t0; h2 = collections.Counter()
for tt, d in ev:
    h2[cur] += tt - last; last = tt; cur += d
print("  large dense kernels at once: " + ", ".join(f"{k}: {100*v/win_ns:.1f} %" for k, v in sorted(h2.items())))
