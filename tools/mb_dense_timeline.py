"""Cycle-counter stamps of one workgroup of the large-tile dense kernel (build: make EXTRA=-DDL_TIMELINE=1 into another .so and
point V3D_HIP_LIB at it).  Prints, for the matrix wave 0 and the loader wave 4 of workgroup 8, the clocks between stamps:
prologue, the first 8 stages, epilogue."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from vision3d_amd import _lib as L  # noqa: E402
from vision3d_amd import synth  # noqa: E402
from vision3d_amd.core.config import second_car_cfg  # noqa: E402
from vision3d_amd.detector import Second  # noqa: E402
from vision3d_amd.runtime import conv2d_split  # noqa: E402


def stamps():
    buf = (C.c_ulonglong * 128)()
    fn = L.lib().v3d_debug_dense_timeline
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
    assert fn(buf) == 0
    return np.array(buf, dtype=np.uint64).reshape(2, 64)


def show(tag):
    torch.cuda.synchronize()
    t = stamps().astype(np.int64)
    for role, name in ((0, "matrix wave 0"), (1, "loader wave 4")):
        r = t[role]
        base = r[0]
        idx = [0, 1, 2] + [3 + i for i in range(16)] + [30, 31, 32, 33]
        vals = [(i, int(r[i] - base)) for i in idx if r[i] > 0]
        print(f"{tag:28s} {name}: " + " ".join(f"{i}:{v}" for i, v in vals))


def main():
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = Second(cfg).cuda().eval().set_precision("bf16x3")  # (the layer-by-layer calls below pass no scale entries: the scale-free arithmetic)
    clouds = [torch.from_numpy(synth.make_cloud(0, 16384)).cuda()]
    with torch.no_grad():
        plan, flat, offsets = model._plan_for(clouds)
        hi, lo = plan.forward_split(flat, offsets)
        occ = plan.bev_occupancy(1).clone()
        dense = model.dense_plan()
        dense.forward(hi, lo)
        bg = dense.background(hi.shape[1], hi.shape[2], hi.device)
        ly = dense.layers[1]
        args = (ly["img"], ly["bias"], ly["relu"], ly["cin"], ly["cout"], ly["k"])
        x_hi, x_lo = bg[0]
        conv2d_split(x_hi, x_lo, *args)
        show("144 px, launch order")
        every = torch.zeros_like(occ)
        work = torch.zeros(2, dtype=torch.int32, device="cuda")
        conv2d_split(x_hi, x_lo, *args, occ=every, reach=1, bg=bg[1], work=work)
        show("80 px persistent, full map")


if __name__ == "__main__":
    main()
