"""Experiment (round 5): one XCD set per frame in flight.

The pipelined SECOND forward keeps 4 frames in flight on 4 HIP streams; every kernel is sized for the whole chip, so the frames'
kernels queue behind each other (round-4 overlap trace: one kernel alone 66 % of the time).  Here each stream is created with
hipExtStreamCreateWithCUMask on a DISJOINT quarter of the chip (2 XCDs = 64 CUs per frame) and the same windows are timed:
  baseline   4 plain streams (what bench.py's pipeline uses)
  layout A   stream q owns mask bits [64 q, 64 q + 64)
  layout B   stream q owns the bits with i % 8 in {2 q, 2 q + 1}
(tools/mb_cumask prints which XCDs a mask reaches, so the layout that means "2 whole XCDs" can be read off its output).
Prints frames/s (median of 15 windows of 20 frames, pipeline empty at the start of a window) and the one-frame-at-a-time latency
on one stream of each kind."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vision3d_amd import synth  # noqa: E402
from vision3d_amd.core import AnchorGenerator  # noqa: E402
from vision3d_amd.core.config import second_car_cfg  # noqa: E402
from vision3d_amd.detector import Second  # noqa: E402


def masked_stream(words):
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def mask_words(bits):
    w = [0] * 8
    for i in bits:
        w[i // 32] |= 1 << (i % 32)
    return w


def windows(run, stream_of_clouds, frames=20, count=15):
    ts = []
    for _ in range(count):
        run.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            run(stream_of_clouds[i % len(stream_of_clouds)])
        run.flush()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return frames / float(np.median(ts))


def main():
    cfg = second_car_cfg()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    frames = [[torch.from_numpy(synth.make_cloud(s, 16384)).cuda()] for s in range(8)]
    layouts = {
        "baseline (plain streams)": None,
        "layout A (64 consecutive bits per stream)": [mask_words(range(64 * q, 64 * q + 64)) for q in range(4)],
        "layout B (bits i % 8 in {2q, 2q+1})": [mask_words([i for i in range(256) if i % 8 in (2 * q, 2 * q + 1)]) for q in range(4)],
        "layout C (128 consecutive bits, 2 frames in flight)": [mask_words(range(128 * q, 128 * q + 128)) for q in range(2)],
    }
    for name, masks in layouts.items():
        torch.manual_seed(0)
        model = Second(cfg).cuda().eval()
        depth = 4 if masks is None else len(masks)
        with torch.no_grad():
            run = model.pipelined_inference(anchors, [16384], depth, autotune=False)
            if masks is not None:
                run.streams = [masked_stream(m) for m in masks]
            for i in range(12):
                run(frames[i % 8])
            run.flush()
            fps = windows(run, frames)
            # one frame at a time on stream 0 of this kind
            g, st = run.slots[0], run.streams[0]
            lat = []
            with torch.cuda.stream(st):
                for i in range(10):
                    g(frames[i % 8])
                torch.cuda.synchronize()
                for i in range(200):
                    t0 = time.perf_counter()
                    g(frames[i % 8])
                    lat.append(time.perf_counter() - t0)
        print(f"{name:52s} depth {depth}: {fps:8.1f} frames/s pipelined, {1e3 * float(np.median(lat)):.3f} ms one frame at a time on one such stream",
              flush=True)
        del run, model
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
