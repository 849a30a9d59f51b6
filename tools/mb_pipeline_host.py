"""Where does a frame's HOST time go in throughput mode?  Times, on the host clock and without waiting for the GPU, the pieces
PipelinedSecond runs per frame: load (copy into the static buffer), graph.replay() and the finalize of collect()."""
import sys
import time

import torch

sys.path.insert(0, ".")
from vision3d_amd import synth  # noqa: E402
from vision3d_amd.core import AnchorGenerator  # noqa: E402
from vision3d_amd.core.config import second_car_cfg  # noqa: E402
from vision3d_amd.detector import Second  # noqa: E402
from vision3d_amd.detector.graph import GraphedSecond  # noqa: E402


def main():
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = Second(cfg).cuda().eval()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    clouds = [torch.from_numpy(synth.make_cloud(0, 16384)).cuda()]
    g = GraphedSecond(model, anchors, [16384])
    with torch.no_grad():
        g.load(clouds)
        g._capture()
        g.graph.replay()
    torch.cuda.synchronize()
    n = 200
    for name, fn in (("load", lambda: g.load(clouds)), ("replay", lambda: g.graph.replay())):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name:8s} host {1e6 * (t1 - t0) / n:8.1f} us per call (enqueue only), {1e6 * (t2 - t0) / n:8.1f} us per call until the GPU is done")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = model.head.finalize_native(*g.outputs, overflow_flag=g.plan.overflow_any())
    t1 = time.perf_counter()
    print(f"finalize host {1e6 * (t1 - t0) / n:8.1f} us per call (GPU idle: the pure host cost of the one 8-byte read + slicing)")
    # replay on 2 / 4 streams from one thread: does the enqueue cost per frame change?
    slots = [GraphedSecond(model, anchors, [16384], slot=i + 1) for i in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    for s, st in zip(slots, streams):
        with torch.cuda.stream(st), torch.no_grad():
            s.load(clouds)
            s._capture()
    torch.cuda.synchronize()
    for k in (1, 2, 4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % k]):
                slots[i % k].graph.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"replay only, {k} stream(s): host {1e6 * (t1 - t0) / n:8.1f} us per frame enqueue, {1e6 * (t2 - t0) / n:8.1f} us per frame total")
    import threading

    def worker(i, k, iters):
        with torch.cuda.stream(streams[i]):
            for _ in range(iters):
                slots[i].graph.replay()
    for k in (2, 4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=worker, args=(i, k, n // k)) for i in range(k)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"replay only, {k} host threads x 1 stream: host {1e6 * (t1 - t0) / n:8.1f} us per frame enqueue, {1e6 * (t2 - t0) / n:8.1f} us per frame total")


if __name__ == "__main__":
    main()
