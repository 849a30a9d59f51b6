"""What do DEAD workgroups cost?  The rulebook / sparse-layer grids are sized by capacities (up to 30x the live counts): every dead
workgroup is dispatched, reads the live count and leaves.  Times v3d_rulebook_subm (hash build + 27-offset table: cap / 256 x 27
workgroups) on ZERO live rows inside one captured graph of 20 calls, for several capacities.
usage (GPU box): python tools/mb_dead_grid.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")
shape = L.host_i32([41, 1600, 1408])
ks = L.host_i32([3, 3, 3])
for cap in (1024, 8192, 20032, 40000, 80000, 160000):
    coords = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    n_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    nbr = torch.empty((27, cap), dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.v3d_rulebook_workspace(cap, cap, 27)), dtype=torch.uint8, device=dev)

    def call():
        L.check(lib.v3d_rulebook_subm(L.ptr(coords), L.ptr(n_dev), cap, shape, ks, L.ptr(nbr), L.ptr(ws), ws.numel(), L.stream_ptr()), "subm")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                call()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100
    blocks = ((cap + 255) // 256) * 27 + min((cap + 255) // 256, 2048)
    print(f"cap {cap:7d}: {blocks:6d} dead workgroups in the call's launches -> {us:7.2f} us per call")
