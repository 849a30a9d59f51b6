"""Development microbenchmarks on the GPU box: back-to-back timings of the library kernels on one real
KITTI-shaped layer (so DVFS / launch gaps do not distort them).  Prints to stdout."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vision3d_amd import synth, _lib as L
from vision3d_amd.core.config import second_car_cfg
from vision3d_amd.spconv import SparseConvTensor
from vision3d_amd.spconv.conv import build_subm_rulebook, build_sparse_rulebook, sparse_conv_forward
from vision3d_amd.spconv.utils import voxelize_batch

def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters, (time.perf_counter() - t0) * 1e6 / iters

cfg = second_car_cfg()
cloud = torch.from_numpy(synth.make_cloud(0)).cuda()
vox, coords, occ, mean, n = voxelize_batch(cloud, [0, cloud.shape[0]], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, 5, 20000)
m = int(n.item())
print("voxels", m)
#print("voxelize_batch gpu/host us: %.1f / %.1f" % timeit(lambda: voxelize_batch(cloud, [0, cloud.shape[0]], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, 5, 20000)))
x = SparseConvTensor(mean[:m], coords[:m], [41, 1600, 1408], 1)
print("subm rulebook gpu/host us: %.1f / %.1f" % timeit(lambda: build_subm_rulebook(x, [3, 3, 3]), 100))
print("sparse rulebook (incl. host sync) gpu/host us: %.1f / %.1f" % timeit(lambda: build_sparse_rulebook(x, [3, 3, 3], [2, 2, 2], [1, 1, 1]), 100))
# walk to stage 2 (64 ch)
rb1 = build_sparse_rulebook(x, [3, 3, 3], [2, 2, 2], [1, 1, 1])
x1 = SparseConvTensor(torch.randn(rb1.n, 32, device="cuda"), rb1.out_indices, rb1.out_shape, 1)
rb2 = build_sparse_rulebook(x1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
x2 = SparseConvTensor(torch.randn(rb2.n, 64, device="cuda"), rb2.out_indices, rb2.out_shape, 1)
rbs = build_subm_rulebook(x2, [3, 3, 3])
w = torch.randn(27, 64, 64, device="cuda") * 0.05
sc, sh = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
pairs = int((rbs.nbr >= 0).sum().item())
print("subm2 layer: n=%d pairs=%d" % (rbs.n, pairs))
import ctypes
raw = ctypes.CDLL(L.LIB_PATH)
REP = 50
def conv_time(feat, wt, rb, s1, s2, algo, flags=0):
    raw.v3d_debug_set_flags(flags); raw.v3d_debug_set_repeat(REP)
    g, h = timeit(lambda: sparse_conv_forward(feat, wt, rb, s1, s2, True, algo), 10, 2)
    raw.v3d_debug_set_flags(0); raw.v3d_debug_set_repeat(1)
    return g / REP
for algo in (1, 2, 3, 4):
    print("conv 64->64 algo %d: %.2f us/launch" % (algo, conv_time(x2.features, w, rbs, sc, sh, algo)))
for flags, what in ((1, "no A loads"), (2, "B always k=0"), (16, "no B loads"), (17, "no A, no B loads"), (4, "no MFMA"), (21, "no A/B/MFMA"), (21+32, "no A/B/MFMA/dsadd"), (64, "no main loop"), (32, "no ds_add only")):
    print("  algo3 ablation %-18s: %.2f us" % (what, conv_time(x2.features, w, rbs, sc, sh, 3, flags)))
x1s = build_subm_rulebook(x1, [3, 3, 3])
w32 = torch.randn(27, 32, 32, device="cuda") * 0.05
for algo in (2, 3, 4):
    print("conv 32->32 (n=%d) algo %d: %.2f us" % (x1s.n, algo, conv_time(x1.features, w32, x1s, None, None, algo)))
xs = build_subm_rulebook(x, [3, 3, 3])
w4 = torch.randn(27, 4, 16, device="cuda")
for algo in (2, 3, 4):
    print("conv 4->16 (n=%d) algo %d: %.2f us" % (xs.n, algo, conv_time(x.features, w4, xs, None, None, algo)))
# NMS at the inference shape
from vision3d_amd.ops.iou_nms import nms_rotated_padded
b = torch.rand(100, 5, device="cuda") * torch.tensor([70, 80, 2, 4, 3.0], device="cuda")
s = torch.rand(100, device="cuda")
print("nms N=100 gpu/host us: %.1f / %.1f" % timeit(lambda: nms_rotated_padded(b, s, 0.01), 200))
d = torch.empty(1, 64, 2, 200, 176, device="cuda")
x4 = SparseConvTensor(torch.randn(3400, 64, device="cuda"), torch.zeros(3400, 4, dtype=torch.int32, device="cuda"), [2, 200, 176], 1)
print("densify gpu/host us: %.1f / %.1f" % timeit(lambda: x4.dense(), 200))
conv = torch.nn.Conv2d(128, 128, 3, padding=1, bias=False).cuda()
inp = torch.randn(1, 128, 200, 176, device="cuda")
with torch.no_grad():
    print("torch conv2d 128->128 3x3 fp32 gpu/host us: %.1f / %.1f" % timeit(lambda: conv(inp), 100))
    convb, inpb = conv.bfloat16(), inp.bfloat16()
    print("torch conv2d 128->128 3x3 bf16 gpu/host us: %.1f / %.1f" % timeit(lambda: convb(inpb), 100))
# dense conv: hand-written bf16x3 MFMA kernel vs MIOpen, GPU-bound loops
from vision3d_amd.runtime import conv2d_split, pack_conv_weight, to_split_nhwc
hi, lo = to_split_nhwc(inp)
img = pack_conv_weight(conv.float().weight)
bias = torch.zeros(128, device="cuda")
lib = L.lib()
yh, yl = torch.empty_like(hi), torch.empty_like(lo)
def dc():
    lib.v3d_conv2d_nhwc_bf16x3(hi.data_ptr(), lo.data_ptr(), img.data_ptr(), bias.data_ptr(), 1, 1, 200, 176, 128, 128, 3,
                               yh.data_ptr(), yl.data_ptr(), 0, L.stream_ptr())
print("v3d conv2d 128->128 3x3 bf16x3 gpu/host us: %.1f / %.1f" % timeit(dc, 200))
