"""Set abstraction at RoI-grid pooling's shape (2 048 keypoints x 512 features, 1 600 grid points, ns = 16 / 32, 516 -> 192 -> 96), device
time per call from 20 calls captured in one HIP graph: the launch-per-layer path against linear_rows + sa_mlp_pair(2), with the fp32
matrix roof (157 TFLOP/s dense v_mfma_f32_16x16x4_f32) beside it.  Run on the GPU box: python tools/mb_sa_pair.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from vision3d_amd import synth
from vision3d_amd.pointnet2 import pointnet2_utils as PU
from vision3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleMSG


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


torch.manual_seed(0)
PEAK = 157.3e12
for name, c, mlp, n, m in (("RoI-grid pooling", 512, [512, 192, 96], 2048, 1600), ("VSA level 3", 64, [64, 64, 64], 10540, 2048)):
    sa = PointnetSAModuleMSG(npoint=-1, radii=[0.8, 1.6] if c == 512 else [1.2, 2.4], nsamples=[16, 32], mlps=[list(mlp), list(mlp)], use_xyz=True).cuda().eval()
    cloud = torch.from_numpy(synth.make_cloud(3)[:, :3]).cuda()[None]
    xyz = cloud[:, PU.furthest_point_sample(cloud, n)[0].long()].contiguous() if n <= 4096 else cloud[:, :n].contiguous()
    new_xyz = (xyz[:, :m] + 0.2).contiguous()
    feat = torch.randn(1, n, c, device="cuda")
    grid = PU.ball_query_grids([(xyz, 2.4)])[0]
    with torch.no_grad():
        nb = PU.ball_query_pair(sa.groupers[0].radius, 16, sa.groupers[1].radius, 32, xyz, new_xyz, grid=grid)
        rows = m * (16 + 32)
        k1, n1, n2 = mlp[0] + 4, mlp[1], mlp[2]
        direct = 2.0 * rows * (k1 * n1 + n1 * n2)
        paired = 2.0 * (n * mlp[0] * 2 * n1 + rows * n1 * n2)
        t_pair2 = graph_us(lambda: sa.fused_forward(xyz, feat, new_xyz, neighbours=nb))
        sa.PAIR_BOTH_SCALES = False
        t_pair = graph_us(lambda: sa.fused_forward(xyz, feat, new_xyz, neighbours=nb))
        sa.PAIR_FIRST_LAYERS = False
        t_layers = graph_us(lambda: sa.fused_forward(xyz, feat, new_xyz, neighbours=nb))
    print(f"{name}: N = {n}, {rows} grouped rows, {k1} -> {n1} -> {n2}")
    print(f"  a launch per layer (4 launches)            {t_layers:7.1f} us   {direct / 1e9:6.2f} GFLOP   {direct / t_layers / 1e6 / PEAK * 1e12:5.2f} of the fp32 matrix peak")
    print(f"  product + pair per scale (3 launches)      {t_pair:7.1f} us   {paired / 1e9:6.2f} GFLOP   {paired / t_pair / 1e6 / PEAK * 1e12:5.2f}")
    print(f"  product + both scales in one (2 launches)  {t_pair2:7.1f} us   {paired / 1e9:6.2f} GFLOP   {paired / t_pair2 / 1e6 / PEAK * 1e12:5.2f}")
