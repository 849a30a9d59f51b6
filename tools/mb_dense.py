"""Timing of the two dense-conv kernels on the RPN shape (128->128 3x3 @ 200x176) with back-to-back launches."""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import _lib as L
from vision3d_amd.runtime import conv2d_split, pack_conv_weight, to_split_nhwc
raw = ctypes.CDLL(L.LIB_PATH)
for (b, h, w, cin, cout, k) in [(1, 200, 176, 128, 128, 3), ]:
    x = torch.randn(b, cin, h, w).cuda()
    wt = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).cuda()
    hi, lo = to_split_nhwc(x)
    img = pack_conv_weight(wt)
    bias = torch.zeros(cout).cuda()
    row = f"{b}x{h}x{w} {cin}->{cout} k{k}: "
    for v in [int(a) for a in sys.argv[1:]] or (1, 2):
        raw.v3d_debug_set_dense_variant(v)
        for _ in range(3):
            conv2d_split(hi, lo, img, bias, True, cin, cout, k, out_split=(cout % 8 == 0), out_nchw=False if cout % 8 == 0 else True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            conv2d_split(hi, lo, img, bias, True, cin, cout, k, out_split=(cout % 8 == 0), out_nchw=False if cout % 8 == 0 else True)
        e1.record(); torch.cuda.synchronize()
        flops = 2.0 * b * h * w * cin * cout * k * k
        t = e0.elapsed_time(e1) / 20 * 1e-3
        row += f" v{v}={t*1e6:7.1f}us ({flops*3/t/1e12:6.1f} TF bf16-equiv)"
    raw.v3d_debug_set_dense_variant(0)
    print(row)

import numpy as np
if not hasattr(raw, 'v3d_debug_dense_timeline'):
    sys.exit(0)
buf = (ctypes.c_ulonglong * 128)()
raw.v3d_debug_set_dense_variant(2)
x = torch.randn(1, 128, 200, 176).cuda(); wt = torch.randn(128, 128, 3, 3).cuda() / 34
hi, lo = to_split_nhwc(x); img = pack_conv_weight(wt); bias = torch.zeros(128).cuda()
for _ in range(3):
    conv2d_split(hi, lo, img, bias, True, 128, 128, 3, out_split=True, out_nchw=False)
torch.cuda.synchronize()
raw.v3d_debug_dense_timeline(buf)
t = np.array(list(buf), dtype=np.int64).reshape(2, 64)
for role in (0, 1):
    r = t[role]; base = t[:, 0].min()
    print("role", role, [int(v - base) for v in r[:20]], "end-of-loop", int(r[30] - base), "epi", [int(r[k] - base) for k in (31, 32, 33)], "in-stage3", [int(r[k] - base) for k in (40, 41, 48, 56, 57)])
