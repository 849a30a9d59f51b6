"""cProfile of the host thread of the PV-RCNN end-to-end loop (two frames in flight, bench.py pvrcnn_end_to_end.run_in_flight): where
the ~1.5 ms of host time per frame go.  Run on the GPU box: python tools/prof_pvrcnn_host.py [frames]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from vision3d_amd import synth
from vision3d_amd.core import AnchorGenerator, Preprocessor
from vision3d_amd.core.config import second_car_cfg
from vision3d_amd.detector import PV_RCNN

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = second_car_cfg()
torch.manual_seed(0)
model = PV_RCNN(cfg).cuda().eval()
pre = Preprocessor(cfg, seed=0)
anchors = AnchorGenerator(cfg).anchors.cuda()
frames = [[torch.from_numpy(synth.make_cloud(j, 16384)).cuda()] for j in range(8)]
AHEAD = 4


def make_item(i):
    return pre(dict(points=frames[i % len(frames)], anchors=anchors))


def run(n):
    out = None
    with torch.no_grad():
        items = model.prefetch_keypoints_many([make_item(j) for j in range(min(AHEAD, n))])
        nxt_j = len(items)
        st = model.inference_begin(items.pop(0), 0)
        prev = None
        for i in range(n):
            if len(items) < AHEAD // 2 + 1 and nxt_j < n:
                batch = [make_item(j) for j in range(nxt_j, min(nxt_j + AHEAD, n))]
                items += model.prefetch_keypoints_many(batch)
                nxt_j += len(batch)
            nxt = model.inference_begin(items.pop(0), (i + 1) % 2) if i + 1 < n else None
            h = model.inference_end(st)
            if prev is not None:
                out = model.inference_collect(prev)
            prev, st = h, nxt
        out = model.inference_collect(prev)
    return out


run(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(n)
torch.cuda.synchronize()
print("plain: %.3f ms per frame" % ((time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
run(n)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
