"""Overlap analysis of a rocprofv3 kernel trace of the pipelined bench: how much of the wall time has 0 / 1 / 2+ kernels running,
and which kernels run alone.  usage: python tools/trace_overlap.py <kernel_trace.csv> [frames_to_skip_fraction]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")) for r in rows]
    ev.sort()
    lo = ev[len(ev) // 2][0]  # second half: the timed, steady-state part of the run
    ev = [e for e in ev if e[0] >= lo]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    points = []
    for s, e, n, q in ev:
        points.append((s, 1, n))
        points.append((e, -1, n))
    points.sort()
    busy = defaultdict(int)
    alone = defaultdict(int)
    active, last = {}, t0
    for t, d, n in points:
        k = sum(active.values())
        busy[min(k, 3)] += t - last
        if k == 1:
            name = next(a for a, c in active.items() if c)
            alone[name] += t - last
        last = t
        active[n] = active.get(n, 0) + d
    wall = t1 - t0
    # one per-frame launch marks a frame: the fill at its start (plan_frame_start_kernel since round 4, v3d_fill + densify before)
    frames = sum(1 for e in ev if e[2].startswith("plan_frame_start")) or sum(1 for e in ev if e[2].startswith("densify_split"))
    print(f"steady-state window {wall / 1e6:.2f} ms, {frames} frames -> {wall / 1e3 / max(frames, 1):.1f} us per frame; queues: {sorted(set(e[3] for e in ev))}")
    for k in sorted(busy):
        print(f"  {k}{'+' if k == 3 else ' '} kernels running: {100.0 * busy[k] / wall:5.1f} % of the time  ({busy[k] / 1e3 / max(frames, 1):6.1f} us per frame)")
    tot = defaultdict(int)
    for s, e, n, q in ev:
        tot[n] += e - s
    print("  kernel                                                         total us/frame   alone us/frame")
    for n in sorted(tot, key=tot.get, reverse=True)[:14]:
        print(f"  {n:62s} {tot[n] / 1e3 / max(frames, 1):8.1f} {alone[n] / 1e3 / max(frames, 1):14.1f}")


if __name__ == "__main__":
    main()
