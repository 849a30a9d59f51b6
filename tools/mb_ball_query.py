"""Ball query, scan kernel vs cell grid (csrc/pointops.hip), device time per call: 20 calls captured in one HIP graph.
Run on the GPU box: python tools/mb_ball_query.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from vision3d_amd import synth
from vision3d_amd.pointnet2 import pointnet2_utils as PU


def graph_us(fn, rep=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(rep):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (10 * rep) * 1e6


for order in ("shuffled", "scan"):
    cloud = torch.from_numpy(synth.make_cloud(5, order=order)[:, :3]).cuda()[None].contiguous()
    kp = cloud[:, PU.furthest_point_sample(cloud, 2048)[0].long()].contiguous()
    for algo in ("scan", "grid"):
        PU.BALL_QUERY_ALGO = algo
        for n, (ra, rb) in ((16384, (0.4, 0.8)), (13000, (0.4, 0.8)), (16384, (0.8, 1.2)), (10000, (1.2, 2.4)), (4600, (2.4, 4.8))):
            db = cloud[:, :n].contiguous()
            print(order, algo, n, ra, rb, "%.1f us" % graph_us(lambda: PU.ball_query_pair(ra, 16, rb, 32, db, kp)), flush=True)
        gp = (kp[:, :1600] + 0.3).contiguous()
        print(order, algo, "roi grid points around 2048 keypoints", "%.1f us" % graph_us(lambda: PU.ball_query_pair(0.8, 16, 1.6, 32, kp, gp)), flush=True)
