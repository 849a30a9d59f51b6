"""A/B of the two arithmetics of the packed sparse kernels on the layers of one real KITTI frame (same process, same box): every
layer's kernel ALONE through the C ABI (v3d_sparse_conv_fwd_packed), REP launches inside one captured HIP graph between two
events.  f16s: the scale entry is computed once outside the timed launches (in the frame it is a table entry, not a launch).
usage: [V3D_HIP_LIB=...] python tools/mb_prec_ab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vision3d_amd.spconv.conv as convmod  # noqa: E402
from vision3d_amd import _lib as L  # noqa: E402
from vision3d_amd import synth  # noqa: E402
from vision3d_amd.core import Preprocessor  # noqa: E402
from vision3d_amd.core.config import second_car_cfg  # noqa: E402
from vision3d_amd.detector import Second  # noqa: E402
from vision3d_amd.runtime import act_entry_from_tensor, rows_split  # noqa: E402

cfg = second_car_cfg()
torch.manual_seed(0)
model = Second(cfg).cuda().eval().set_precision("bf16x3")  # (only to walk the layers: the timed calls below choose the arithmetic)
cloud = torch.from_numpy(synth.make_cloud(0, 16384)).cuda()
orig = convmod.sparse_conv_forward
cap = []
convmod.sparse_conv_forward = lambda *a, **k: (cap.append(a), orig(*a, **k))[1]
with torch.no_grad():
    it = Preprocessor(cfg, seed=0)(dict(points=[cloud]))
    model.cnn(it["voxel_mean"], it["coordinates"], it["batch_size"])
convmod.sparse_conv_forward = orig
REP = 25


def timed(launch):
    g = torch.cuda.CUDAGraph()
    launch()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(REP):
            launch()
    ts = []
    for trial in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        if trial:
            ts.append(e0.elapsed_time(e1) * 1e3 / REP)
    return float(np.median(ts))


tot = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
print(f"lib: {L.LIB_PATH}")
print("columns: fp32 rows in / out (in-register split) | +pre = gathered rows pre-split, output rows written only pre-split (a throughput-mode plan)")
for a in cap:
    feat, w, rb, sc, sh, relu = a[0], a[1], a[2], a[3], a[4], a[5]
    cin, cout = w.shape[-2], w.shape[-1]
    if cin < 16:
        continue
    k = rb.nbr.shape[0]
    flat = w.reshape(-1, cin, cout).contiguous()
    row = f"{cin:3d}->{cout:3d} K={k:2d} n={rb.n:6d}"
    for prec, name in ((0, "bf16x3"), (1, "f16s")):
        pname = "fp32" if prec else "bf16x3"
        img = convmod.pack_sparse_weight(flat, k, cin, cout, pname)
        entry = act_entry_from_tensor(feat) if prec else None
        out = torch.empty((rb.n, cout), dtype=torch.float32, device="cuda")

        def launch(img=img, entry=entry, out=out, prec=prec):
            L.check(L.lib().v3d_sparse_conv_fwd_packed(L.ptr(feat), L.ptr(img), L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap, k, cin, cout,
                                                        L.ptr(sc), L.ptr(sh), int(bool(relu)), L.ptr(out), int(rb.n), prec,
                                                        L.ptr(entry), None, None, None, None, L.stream_ptr()), "fwd")
        t = timed(launch)
        tot[prec] += t
        row += f"  {name}={t:6.2f}us"
        in_s = rows_split(feat.contiguous(), pname, entry)
        out_s = torch.empty((rb.n, 2 * cout), dtype=torch.int16, device="cuda")
        nxt = act_entry_from_tensor(out) if prec else None

        def launch_pre(img=img, entry=entry, prec=prec, in_s=in_s, out_s=out_s, nxt=nxt):
            L.check(L.lib().v3d_sparse_conv_fwd_packed(None, L.ptr(img), L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap, k, cin, cout,
                                                        L.ptr(sc), L.ptr(sh), int(bool(relu)), None, int(rb.n), prec,
                                                        L.ptr(entry), L.ptr(nxt), None, L.ptr(in_s), L.ptr(out_s), L.stream_ptr()), "fwd")
        t = timed(launch_pre)
        tot[2 + prec] += t
        row += f" +pre={t:6.2f}us"
    print(row)
print(f"sum: bf16x3 {tot[0]:.1f} us (+pre {tot[2]:.1f}), f16s {tot[1]:.1f} us (+pre {tot[3]:.1f})")
