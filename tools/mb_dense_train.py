"""Cycle-counter timeline of the 3x3 training convolution (csrc/dense_train.hip built with -DDT_TIMELINE=1 as a second library:
make -C vision3d_amd/csrc EXTRA=-DDT_TIMELINE=1, copy to lib_tl.so, V3D_HIP_LIB=...) and isolated timings of the dense-train kernels.
usage: python tools/mb_dense_train.py [B H W]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import _lib as L
lib = L.lib()
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 200, 176)
x = torch.randn(B, 128, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(128, 128, 3, 3, device="cuda") / 34.0
img = torch.empty(int(lib.v3d_dense_train_weight_image_bytes(3)), dtype=torch.uint8, device="cuda")
L.check(lib.v3d_dense_train_pack_weights(L.ptr(w), 3, 0, L.ptr(img), L.stream_ptr()), "pack")
y = torch.empty_like(x)
stats = torch.empty((lib.v3d_dense_train_conv_tiles(B, H, W), 2, 128), device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


conv = lambda: L.check(lib.v3d_dense_train_conv(L.ptr(x), L.ptr(img), B, H, W, 3, L.ptr(y), L.ptr(stats), L.stream_ptr()), "conv")
print(f"conv 3x3 ({B},{H},{W}): {timed(conv):.1f} us   (MFMA floor {2.0 * B * H * W * 128 * 128 * 9 / 2.5e15 * 1e6:.1f} us at 2.5 PF)")
ws = torch.empty(int(lib.v3d_dense_train_wgrad_workspace(3)), dtype=torch.uint8, device="cuda")
dw = torch.empty((128, 128, 3, 3), device="cuda")
wg2 = lambda: L.check(lib.v3d_dense_train_wgrad(L.ptr(x), L.ptr(y), B, H, W, 3, L.ptr(dw), L.ptr(ws), ws.numel(), L.stream_ptr()), "wgrad")
print(f"wgrad 3x3 (+ reduce): {timed(wg2):.1f} us")
try:
    fn = C.CDLL(L.LIB_PATH).v3d_debug_dense_train_timeline
except AttributeError:
    sys.exit(0)
conv(); torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
fn(buf)
t = np.array(buf, dtype=np.int64).reshape(2, 128)
t0 = t[t > 0].min()
for role, name in ((0, "multiplier wave 0: stage | top | first-half MFMAs issued | fragment reads returned | barrier passed"),
                   (1, "loader wave 8: stage | before wait | landed | barrier passed | requests issued")):
    print(name)
    for g in range(28):
        r = t[role, 4 * g:4 * g + 4]
        if r.min() <= 0:
            continue
        nxt = t[role, 4 * g + 4] if 4 * g + 4 < 128 else 0
        print(f"  {g:3d} | " + " | ".join(f"{int(v - t0):7d}" for v in r) + f" | stage total {int(nxt - r[0]) if nxt > 0 else -1}")
print("tile epilogues of multiplier wave 0 (start, end, clocks):")
for k in range(8):
    a, b = t[0, 112 + 2 * k], t[0, 113 + 2 * k]
    if a > 0 and b > 0:
        print(f"  tile {k}: {int(a - t0):8d} {int(b - t0):8d} {int(b - a):7d}")
