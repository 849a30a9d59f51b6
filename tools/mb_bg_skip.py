"""Background skipping of the dense head: kernel time of one 3x3 128->128 convolution at 200x176 with (a) no bitmap, (b) an empty
map's bitmap (every tile background), (c) the bitmap of a real frame at reach 1..6 -- and the tile fractions that go with it."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from vision3d_amd import synth  # noqa: E402
from vision3d_amd.core.config import second_car_cfg  # noqa: E402
from vision3d_amd.detector import Second  # noqa: E402
from vision3d_amd.runtime import conv2d_split  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


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
        print(f"no bitmap            : {timed(lambda: conv2d_split(x_hi, x_lo, *args)):7.1f} us")
        none = torch.full_like(occ, -1)
        print(f"empty map (all skip) : {timed(lambda: conv2d_split(x_hi, x_lo, *args, occ=none, reach=1, bg=bg[1])):7.1f} us")
        every = torch.zeros_like(occ)
        print(f"full map (no skip)   : {timed(lambda: conv2d_split(x_hi, x_lo, *args, occ=every, reach=1, bg=bg[1])):7.1f} us")
        work = torch.zeros(2, dtype=torch.int32, device="cuda")
        for name, w in (("144-px tiles in launch order (no counter)", None), ("80-px tiles, persistent grid (counter)", work)):
            for reach in (1, 6):
                t = timed(lambda: conv2d_split(x_hi, x_lo, *args, occ=occ, reach=reach, bg=bg[1], work=w))
                t0 = timed(lambda: conv2d_split(x_hi, x_lo, *args, occ=every, reach=reach, bg=bg[1], work=w))
                t1 = timed(lambda: conv2d_split(x_hi, x_lo, *args, occ=none, reach=reach, bg=bg[1], work=w))
                print(f"{name:42s} reach {reach}: real frame {t:7.1f} us   full map {t0:7.1f} us   empty map {t1:7.1f} us")

if __name__ == "__main__":
    main()
