"""In-frame timeline of the one-frame-at-a-time graph from a rocprofv3 kernel trace: for every POSITION of the frame's kernel
sequence the kernel, its duration (median / p10 / p90 over the steady-state frames) and the gap between the previous kernel's end
and its start.  Answers "which launches of a kernel are the slow ones in the frame" (the per-kernel averages of --stats cannot).
usage: python tools/trace_sequence.py <kernel_trace.csv> [frames_to_use=200]
       (trace: rocprofv3 --kernel-trace --output-format csv -- python bench.py --pipeline 1 --steps 300 --no-cpu-baseline --no-roofline --no-h2d --windows 1)"""
import csv
import sys

import numpy as np


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    want = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    # a frame starts at the kernel in front of vox_insert: its per-frame fill (v3d_fill_kernel / plan_frame_start_kernel), the
    # graph's first node
    starts = [i for i in range(len(ev) - 1) if ev[i + 1][2].startswith("vox_insert") and "fill" in ev[i][2] + "fill" * ev[i][2].startswith("plan_frame_start")]
    frames = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]
    if not frames:
        print("no frames found")
        return
    length = int(np.median([len(f) for f in frames]))
    frames = [f for f in frames if len(f) == length][-want:]
    # eager kernels between two graphs (cloud copy, finalize) are part of the sequence: keep whatever repeats
    names = [frames[-1][i][2] for i in range(length)]
    frames = [f for f in frames if all(f[i][2] == names[i] for i in range(length))]
    dur = np.array([[f[i][1] - f[i][0] for i in range(length)] for f in frames]) / 1e3
    gap = np.array([[f[i][0] - f[i - 1][1] if i else 0 for i in range(length)] for f in frames]) / 1e3
    span = np.array([f[-1][1] - f[0][0] for f in frames]) / 1e3
    print(f"{len(frames)} frames of {length} kernels; first start -> last end {np.median(span):.1f} us median; "
          f"sum of kernel durations {np.median(dur.sum(1)):.1f} us, sum of gaps {np.median(gap.sum(1)):.1f} us")
    print(f"{'#':>3s} {'kernel':70s} {'dur med':>8s} {'p10':>7s} {'p90':>7s} {'gap med':>8s}")
    for i, n in enumerate(names):
        short = n.split("(")[0].replace("void ", "")[:70]
        print(f"{i:3d} {short:70s} {np.median(dur[:, i]):8.2f} {np.percentile(dur[:, i], 10):7.2f} {np.percentile(dur[:, i], 90):7.2f} "
              f"{np.median(gap[:, i]):8.2f}")


if __name__ == "__main__":
    main()
