"""GPU-bound timing of the small kernels: raw ctypes calls in a tight loop (host cost << kernel time)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision3d_amd import synth, _lib as L
from vision3d_amd.core.config import second_car_cfg
cfg = second_car_cfg()
lib = L.lib()
cloud = torch.from_numpy(synth.make_cloud(0)).cuda()
n = cloud.shape[0]
coords = torch.empty((n, 4), dtype=torch.int32, device="cuda"); occ = torch.empty(n, dtype=torch.int32, device="cuda")
mean = torch.empty((n, 4), device="cuda"); nv = torch.zeros(1, dtype=torch.int32, device="cuda")
ws = L.workspace(lib.v3d_voxelize_workspace(n), "cuda")
offs = L.host_i32([0, n]); vs = L.host_f32(cfg.VOXEL_SIZE); bd = L.host_f32(cfg.GRID_BOUNDS)
st = L.stream_ptr()
def vox():
    lib.v3d_voxelize(cloud.data_ptr(), n, 4, offs, 1, vs, bd, 5, 20000, 0, coords.data_ptr(), occ.data_ptr(), mean.data_ptr(), nv.data_ptr(), ws.data_ptr(), ws.numel(), st)
def timeit(fn, iters=300, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters): fn()
    e1.record(); host = (time.perf_counter() - t0) * 1e6 / iters
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters, host
print("voxelize (4 kernels + memset) gpu/host-enqueue us: %.1f / %.1f" % timeit(vox))
m = int(nv.item())
shape = L.host_i32([41, 1600, 1408]); ks = L.host_i32([3, 3, 3])
nbr = torch.empty((27, n), dtype=torch.int32, device="cuda")
ws2 = L.workspace(lib.v3d_rulebook_workspace(n, n, 27), "cuda")
def subm():
    lib.v3d_rulebook_subm(coords.data_ptr(), nv.data_ptr(), n, shape, ks, nbr.data_ptr(), ws2.data_ptr(), ws2.numel(), st)
print("subm rulebook (memset + hash_build + nbr) gpu/host us: %.1f / %.1f" % timeit(subm))
co = torch.empty((2 * n, 4), dtype=torch.int32, device="cuda"); no = torch.zeros(1, dtype=torch.int32, device="cuda")
nbr2 = torch.empty((27, 2 * n), dtype=torch.int32, device="cuda"); ov = torch.zeros(1, dtype=torch.int32, device="cuda")
ws3 = L.workspace(lib.v3d_rulebook_workspace(n, 2 * n, 27), "cuda")
stv = L.host_i32([2, 2, 2]); pdv = L.host_i32([1, 1, 1])
def sparse():
    lib.v3d_rulebook_sparse(coords.data_ptr(), nv.data_ptr(), n, shape, ks, stv, pdv, co.data_ptr(), no.data_ptr(), 2 * n, nbr2.data_ptr(), ov.data_ptr(), ws3.data_ptr(), ws3.numel(), st)
print("sparse rulebook (3 memsets + 5 kernels) gpu/host us: %.1f / %.1f" % timeit(sparse))
b = torch.rand(100, 5, device="cuda") * torch.tensor([70, 80, 2, 4, 3.0], device="cuda"); s = torch.rand(100, device="cuda")
keep = torch.empty(100, dtype=torch.int64, device="cuda"); nk = torch.zeros(1, dtype=torch.int32, device="cuda")
ws4 = L.workspace(lib.v3d_nms_rotated_workspace(100), "cuda")
def nms():
    lib.v3d_nms_rotated(b.data_ptr(), s.data_ptr(), 100, C.c_float(0.01), keep.data_ptr(), nk.data_ptr(), ws4.data_ptr(), ws4.numel(), st)
print("nms N=100 (5 kernels) gpu/host us: %.1f / %.1f" % timeit(nms))
