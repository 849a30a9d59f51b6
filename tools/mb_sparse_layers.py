"""Per-layer timing of the packed sparse product (algo 4) on the layers of one real frame: the automatic kernel choice and
each forced kernel (1 = 16-row, 10 = LDS ring, 5 = 64-row LDS-shared weights; a negative rows_hint at the C ABI).  Every
launch is repeated REP times inside a captured HIP graph between two HIP events: the kernel's own duration, no interpreter time.
usage: python tools/mb_sparse_layers.py [kitti|waymo] [batch]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import synth
from vision3d_amd.core import Preprocessor
from vision3d_amd.core.config import second_car_cfg, waymo_range_cfg
from vision3d_amd.detector import Second
import vision3d_amd.spconv.conv as convmod

wl = sys.argv[1] if len(sys.argv) > 1 else "kitti"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = waymo_range_cfg() if wl == "waymo" else second_car_cfg()
torch.manual_seed(0)
model = Second(cfg).cuda().eval()
ORDER = os.environ.get("MB_ORDER", "shuffled")  # "scan": points in firing order (the inference dataset), see synth.make_cloud
mk = (lambda s: synth.make_waymo_cloud(s, 180000, order=ORDER)) if wl == "waymo" else (lambda s: synth.make_cloud(s, 16384, order=ORDER))
clouds = [torch.from_numpy(mk(i)).cuda() for i in range(bs)]
orig = convmod.sparse_conv_forward
cap = []
convmod.sparse_conv_forward = lambda *a, **k: (cap.append(a), orig(*a, **k))[1]
with torch.no_grad():
    it = Preprocessor(cfg, seed=0)(dict(points=clouds))
    model.cnn(it["voxel_mean"], it["coordinates"], it["batch_size"])
convmod.sparse_conv_forward = orig
REP = 25
PREC = os.environ.get("MB_PRECISION", "bf16x3")  # arithmetic of the timed calls ("fp32" = f16s: each call then also launches the
# small scale-entry reduction of the op-by-op path -- tools/mb_prec_ab.py times the kernels alone in both arithmetics)
VARIANTS = [int(v) for v in os.environ.get("MB_VARIANTS", "0,1,10,5").split(",")]  # forced kernels (negative rows_hint codes)


def packed_for(a):
    w = a[1]
    cin, cout = w.shape[-2], w.shape[-1]
    return convmod.pack_sparse_weight(w.reshape(-1, cin, cout).contiguous(), a[2].nbr.shape[0], cin, cout, PREC)


def timed(a, variant):
    g = torch.cuda.CUDAGraph()
    img = packed_for(a)
    with torch.no_grad():
        orig(a[0], a[1], a[2], a[3], a[4], a[5], 4, img, variant, PREC)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(REP):
                o = orig(a[0], a[1], a[2], a[3], a[4], a[5], 4, img, variant, PREC)
    ts = []
    for trial in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        if trial: ts.append(e0.elapsed_time(e1) * 1e3 / REP)
    return float(np.mean(ts)), o


print(f"{wl} bs={bs} order={ORDER}")
for a in cap:
    cin, cout = a[1].shape[-2], a[1].shape[-1]
    if cin < 16: continue
    rb = a[2]
    row = f"{cin:3d}->{cout:3d} K={rb.nbr.shape[0]:2d} n={rb.n:6d}"
    ref = None
    for v in VARIANTS:
        t, o = timed(a, v)
        ref = o.clone() if ref is None else ref
        assert (ref - o).abs().max() <= 1e-4 * ref.abs().max()
        row += f"  {'auto' if v == 0 else 'v' + str(v)}={t:7.1f}us"
    print(row)

# the same layers launched in FRAME ORDER (one after the other, REP frames in a graph): what the sequence costs when every
# layer finds the caches as the previous layers left them, against the sum of the isolated (hot) timings above
if os.environ.get("MB_SEQUENCE", "1") != "0":
    layers = [a for a in cap if a[1].shape[-2] >= 16]
    g = torch.cuda.CUDAGraph()
    with torch.no_grad():
        imgs = [packed_for(a) for a in layers]
        for a, img in zip(layers, imgs):
            orig(a[0], a[1], a[2], a[3], a[4], a[5], 4, img, 0, PREC)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(REP):
                for a, img in zip(layers, imgs):
                    orig(a[0], a[1], a[2], a[3], a[4], a[5], 4, img, 0, PREC)
    ts = []
    for trial in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        if trial: ts.append(e0.elapsed_time(e1) * 1e3 / REP)
    iso = sum(timed(a, 0)[0] for a in layers)
    print(f"frame order: {float(np.mean(ts)):7.1f} us per {len(layers)}-layer sequence; sum of the isolated timings {iso:7.1f} us")
    # with 64 MB of unrelated traffic between the sequences (another frame's dense head in flight)
    junk = torch.empty(16 * 1024 * 1024, device="cuda")
    g2 = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g2):
        for _ in range(REP):
            junk.add_(1.0)
            for a in layers:
                orig(a[0], a[1], a[2], a[3], a[4], a[5], 4, a[7], 0)
    g3 = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g3):
        for _ in range(REP):
            junk.add_(1.0)
    def tg(gr):
        ts = []
        for trial in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            if trial: ts.append(e0.elapsed_time(e1) * 1e3 / REP)
        return float(np.mean(ts))
    print(f"frame order behind a 128 MB stream: {tg(g2) - tg(g3):7.1f} us per sequence")

# Does a launch cost more when its inputs were JUST WRITTEN by the previous kernel (as in a frame) than when they are old?
# [rewrite features; layer] x REP minus [rewrite features] x REP, against [layer] x REP; the same for the neighbour table.
if os.environ.get("MB_FRESH", "1") != "0":
    a = [x for x in cap if x[1].shape[-2] == 64 and x[1].shape[-1] == 64 and x[2].nbr.shape[0] == 27][0]
    feat_src, nbr_src = a[0].clone(), a[2].nbr.clone()

    def tgraph(fn):
        g = torch.cuda.CUDAGraph()
        with torch.no_grad():
            fn(); torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for _ in range(REP):
                    fn()
        ts = []
        for trial in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            if trial: ts.append(e0.elapsed_time(e1) * 1e3 / REP)
        return float(np.mean(ts))

    layer = lambda: orig(a[0], a[1], a[2], a[3], a[4], a[5], 4, a[7], 0)
    wf = lambda: a[0].mul_(1.0)          # rewrites the feature rows in place (same values)
    wn = lambda: a[2].nbr.add_(0)        # rewrites the neighbour table in place
    t_layer, t_wf, t_wn = tgraph(layer), tgraph(wf), tgraph(wn)
    t_f = tgraph(lambda: (wf(), layer())) - t_wf
    t_n = tgraph(lambda: (wn(), layer())) - t_wn
    t_fn = tgraph(lambda: (wf(), wn(), layer())) - t_wf - t_wn
    print(f"64->64 n={a[2].n}: inputs old {t_layer:6.1f} us | features just rewritten {t_f:6.1f} | neighbour table just rewritten {t_n:6.1f} | both {t_fn:6.1f}")
