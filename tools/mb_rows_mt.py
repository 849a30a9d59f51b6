"""Per-layer timing of the packed sparse kernel with 1 / 2 / 4 row tiles per workgroup (REP back-to-back launches
inside one HIP-event bracket).  usage: python tools/mb_rows_mt.py [kitti|waymo] [batch]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import synth, _lib as L
from vision3d_amd.core import AnchorGenerator, Preprocessor
from vision3d_amd.core.config import second_car_cfg, waymo_range_cfg
from vision3d_amd.detector import Second
import vision3d_amd.spconv.conv as convmod

wl = sys.argv[1] if len(sys.argv) > 1 else "kitti"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = waymo_range_cfg() if wl == "waymo" else second_car_cfg()
torch.manual_seed(0)
model = Second(cfg).cuda().eval()
pre = Preprocessor(cfg, seed=0)
mk = (lambda s: synth.make_waymo_cloud(s, 180000)) if wl == "waymo" else (lambda s: synth.make_cloud(s, 16384))
clouds = [torch.from_numpy(mk(i)).cuda() for i in range(bs)]
raw = ctypes.CDLL(L.LIB_PATH)
orig = convmod.sparse_conv_forward
cap = []
def capture(*a, **k):
    cap.append((a, k)); return orig(*a, **k)
convmod.sparse_conv_forward = capture
with torch.no_grad():
    acfg = cfg
    if wl == "waymo":
        acfg = cfg.clone()
        acfg.GRID_BOUNDS = [cfg.GRID_BOUNDS[0], cfg.GRID_BOUNDS[1], cfg.GRID_BOUNDS[2], cfg.GRID_BOUNDS[3] + 0.02,
                            cfg.GRID_BOUNDS[4] + 0.02, cfg.GRID_BOUNDS[5]]
    it = pre(dict(points=clouds, anchors=AnchorGenerator(acfg).anchors.cuda()))
    model.cnn(it["voxel_mean"], it["coordinates"], it["batch_size"])  # the per-op backbone (inference(item) runs the fused plan)
convmod.sparse_conv_forward = orig
REP = 25
print(f"{wl} bs={bs}")
for (a, k) in cap:
    features, weight, rb = a[0], a[1], a[2]
    cin, cout = weight.shape[-2], weight.shape[-1]
    if cin < 32: continue
    row = f"{cin:3d}->{cout:3d} K={rb.nbr.shape[0]:2d} n={rb.n:6d} cap={rb.cap:6d} "
    ref = None
    for mt in (1, 10, 11, 5):
        raw.v3d_debug_set_rows_mt(mt)
        ts = []
        for trial in range(4):
            raw.v3d_debug_set_repeat(REP)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o = orig(*a, **k); e1.record()
            raw.v3d_debug_set_repeat(1); torch.cuda.synchronize()
            if trial: ts.append(e0.elapsed_time(e1) * 1e3 / REP)
        if ref is None: ref = o.clone()
        elif os.environ.get("MB_NOCHECK"): pass
        else: assert torch.equal(ref, o) or (ref - o).abs().max() < 1e-3 * ref.abs().max(), "mt variants disagree"
        row += f" mt{mt}={np.mean(ts):7.1f}us" + (f" (maxdiff/max {float((ref - o).abs().max() / ref.abs().max()):.1e})" if mt != 1 else "")
    raw.v3d_debug_set_rows_mt(0)
    print(row)
