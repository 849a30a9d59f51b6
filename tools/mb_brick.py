"""Round 6: the brick kernel (csrc/brick.hip) against the offset-outer kernel on the submanifold layers of one real frame, rows in
first-touch order of a SHUFFLED sweep (no locality) and of a host-side Morton-sorted sweep (spatial order: what a plan in brick
order produces).  Per layer: the planner's launch time, the statistics of its tables, both kernels' times (REP launches inside a
captured graph between two HIP events) and whether the outputs are the same bits.
usage: python tools/mb_brick.py [waymo|kitti] [batch]        (MB_PRECISION=fp32|bf16x3)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import synth, _lib as L
from vision3d_amd.core import Preprocessor
from vision3d_amd.core.config import second_car_cfg, waymo_range_cfg
from vision3d_amd.detector import Second
import vision3d_amd.spconv.conv as convmod

wl = sys.argv[1] if len(sys.argv) > 1 else "waymo"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
PREC = os.environ.get("MB_PRECISION", "fp32")
REP = 20
cfg = waymo_range_cfg() if wl == "waymo" else second_car_cfg()
lib = L.lib()


def graph_time(fn):
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


def brick_tables(rb, K):
    cap = rb.cap
    sz = [ctypes.c_size_t() for _ in range(4)]
    lib.v3d_sparse_brick_table_bytes(cap, K, *[ctypes.byref(s) for s in sz])
    t = [torch.zeros(int(s.value), dtype=torch.uint8, device="cuda") for s in sz]
    return t


def run_layers(order):
    torch.manual_seed(0)
    model = Second(cfg).cuda().eval()
    mk = (lambda s: synth.make_waymo_cloud(s, 180000, order=order)) if wl == "waymo" else (lambda s: synth.make_cloud(s, 16384, order=order))
    clouds = [torch.from_numpy(mk(i)).cuda() for i in range(bs)]
    orig = convmod.sparse_conv_forward
    cap = []
    convmod.sparse_conv_forward = lambda *a, **k: (cap.append(a), orig(*a, **k))[1]
    with torch.no_grad():
        it = Preprocessor(cfg, seed=0)(dict(points=clouds))
        model.cnn(it["voxel_mean"], it["coordinates"], it["batch_size"])
    convmod.sparse_conv_forward = orig
    print(f"== {wl} bs={bs} order={order} precision={PREC}")
    prec = L.PRECISIONS[PREC]
    seen = {}
    for a in cap:
        feat, w, rb = a[0], a[1], a[2]
        cin, cout = w.shape[-2], w.shape[-1]
        K = rb.nbr.shape[0]
        if K != 27 or (cin, cout) not in ((64, 64), (32, 32)) or rb.n != feat.shape[0]:
            continue  # submanifold 3x3x3 layers of the shapes the brick kernel covers
        key = (rb.nbr.data_ptr(), cin)
        if key in seen:
            continue
        seen[key] = 1
        n, capr = rb.n, rb.cap
        wf = w.reshape(-1, cin, cout).contiguous().float()
        img = convmod.pack_sparse_weight(wf, K, cin, cout, PREC)
        sc = torch.rand(cout, device="cuda") + 0.5
        sh = torch.randn(cout, device="cuda") * 0.1
        ent_in = torch.empty(4, device="cuda"); ent_next = torch.tensor([1.0, 1.0, 32768.0, 0.0], device="cuda")
        flag = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        L.check(lib.v3d_act_scale_from_rows(L.ptr(feat), None, feat.shape[0], cin, 0, L.ptr(ent_in), None, L.stream_ptr()), "scale")
        fsplit = torch.empty((capr, 2 * cin), dtype=torch.int16, device="cuda")
        L.check(lib.v3d_sparse_rows_split(L.ptr(feat), L.ptr(rb.n_dev), feat.shape[0], cin, prec, L.ptr(ent_in), L.ptr(fsplit), L.stream_ptr()), "split")
        tabs = brick_tables(rb, K)
        plan = lambda: L.check(lib.v3d_sparse_brick_plan(L.ptr(rb.nbr), L.ptr(rb.n_dev), capr, K, *[L.ptr(t) for t in tabs], L.stream_ptr()), "brick_plan")
        plan(); torch.cuda.synchronize()
        ucnt = tabs[2].view(torch.int32)[: (n + 255) // 256].cpu().numpy()
        tm = tabs[3].view(torch.int32)[: (n + 15) // 16].cpu().numpy().astype(np.uint32)
        tile_frac = np.unpackbits(tm.view(np.uint8)).sum() / (27.0 * len(tm))
        out_a = torch.zeros((capr, cout), device="cuda"); out_b = torch.zeros((capr, cout), device="cuda")
        os_a = torch.zeros((capr, 2 * cout), dtype=torch.int16, device="cuda"); os_b = torch.zeros_like(os_a)

        def old(variant, out=out_a, osp=os_a):
            L.check(lib.v3d_sparse_conv_fwd_packed(None, L.ptr(img), L.ptr(rb.nbr), L.ptr(rb.n_dev), capr, K, cin, cout, L.ptr(sc), L.ptr(sh), 1,
                                                    L.ptr(out), -variant if variant else n, prec, L.ptr(ent_in), L.ptr(ent_next), L.ptr(flag),
                                                    L.ptr(fsplit), L.ptr(osp), L.stream_ptr()), "packed2")

        def new(out=out_b, osp=os_b):
            L.check(lib.v3d_sparse_conv_fwd_brick(L.ptr(fsplit), L.ptr(img), L.ptr(rb.nbr), *[L.ptr(t) for t in tabs], L.ptr(rb.n_dev), capr, K, cin, cout,
                                                  L.ptr(sc), L.ptr(sh), 1, L.ptr(out), prec, L.ptr(ent_in), L.ptr(ent_next), L.ptr(flag), L.ptr(osp),
                                                  L.stream_ptr()), "brick")

        ref_variant = 6 if (cin, cout) == (64, 64) else 5   # offset-outer / 64-row kernel: every wave walks the offsets in order
        old(ref_variant); new(); torch.cuda.synchronize()
        same = bool(torch.equal(out_a[:n], out_b[:n]) and torch.equal(os_a[:n], os_b[:n]))
        maxdiff = float((out_a[:n] - out_b[:n]).abs().max())
        t_plan = graph_time(plan)
        t_auto = graph_time(lambda: old(0))
        t_ref = graph_time(lambda: old(ref_variant))
        t_new = graph_time(new)
        t_new_split_only = graph_time(lambda: new(out=None))
        print(f"{cin:3d}->{cout:3d} n={n:6d} cap={capr:6d} | union/256 mean {ucnt.mean() / 256:.2f} max {ucnt.max() / 256:.2f} overflow passes {(ucnt > 480).sum()}/{len(ucnt)} "
              f"tile-offset frac {tile_frac:.2f} | plan {t_plan:6.1f} us | auto {t_auto:6.1f}  v{ref_variant} {t_ref:6.1f}  brick {t_new:6.1f} (split rows only {t_new_split_only:6.1f}) us | "
              f"same bits as v{ref_variant}: {same} (max diff {maxdiff:.3g})")


for order in os.environ.get("MB_ORDERS", "morton,shuffled").split(","):
    run_layers(order)
