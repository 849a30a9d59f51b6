"""Hand-written dense TRAIN path (csrc/dense_train.hip, vision3d_amd/dense_train.py) against torch.

Bar: the kernels store activations and gradients in bf16 and accumulate in fp32 -- the arithmetic of the same modules under
`torch.autocast(bfloat16)`.  Each kernel is checked against a float32/float64 torch statement of the same operation fed with
the SAME bf16-rounded inputs (so the only differences are the fp32 accumulation order and the final bf16 rounding: tolerance a
few bf16 ulps, 2^-8 relative), and the whole stack (7 x conv + BatchNorm + ReLU, heads, loss-side gradient) against the fp32
torch modules within bf16 tolerance, plus bit-repeatability.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16_EPS = 2.0 ** -8


def _lib():
    from vision3d_amd import _lib as L
    return L, L.lib()


def _bf16_nhwc(x):
    """fp32 (B, C, H, W) -> bf16 channels_last tensor (its storage IS the (B, H, W, C) array the kernels take)."""
    return x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("ksize", [3, 1])
@pytest.mark.parametrize("shape", [(1, 24, 20), (2, 33, 16), (1, 200, 176)])
def test_conv_forward_and_data_gradient_form(ksize, shape):
    """y = conv(x, W) incl. the per-tile channel statistics; and the SAME kernel on the transposed / tap-flipped image = the data
    gradient (conv_transpose)."""
    L, lib = _lib()
    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(ksize * 100 + H)
    x = _bf16_nhwc(torch.randn(B, 128, H, W, device="cuda", generator=g))
    w = (torch.randn(128, 128, ksize, ksize, device="cuda", generator=g) / (128 * ksize * ksize) ** 0.5).contiguous()
    wb = w.to(torch.bfloat16).float()
    tiles = lib.v3d_dense_train_conv_tiles(B, H, W)
    for transpose in (0, 1):
        img = torch.empty(int(lib.v3d_dense_train_weight_image_bytes(ksize)), dtype=torch.uint8, device="cuda")
        L.check(lib.v3d_dense_train_pack_weights(L.ptr(w), ksize, transpose, L.ptr(img), L.stream_ptr()), "pack")
        y = torch.empty_like(x)
        stats = torch.full((tiles, 2, 128), float("nan"), device="cuda")
        L.check(lib.v3d_dense_train_conv(L.ptr(x), L.ptr(img), B, H, W, ksize, L.ptr(y), L.ptr(stats), L.stream_ptr()), "conv")
        ref = (F.conv2d(x.float(), wb, padding=ksize // 2) if not transpose
               else F.conv_transpose2d(x.float(), wb, padding=ksize // 2))
        assert _rel(y.float(), ref) < 2 * BF16_EPS, (ksize, shape, transpose, _rel(y.float(), ref))
        # statistics = sums over the ROUNDED outputs, tile partials in tile order
        s = stats.double().sum(0)
        yy = y.float().double()
        assert torch.allclose(s[0], yy.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
        assert torch.allclose(s[1], (yy * yy).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
        y2 = torch.empty_like(x)
        L.check(lib.v3d_dense_train_conv(L.ptr(x), L.ptr(img), B, H, W, ksize, L.ptr(y2), 0, L.stream_ptr()), "conv")
        assert torch.equal(y, y2)  # repeatable, and the statistics output is optional


@pytest.mark.parametrize("ksize", [3, 1])
@pytest.mark.parametrize("shape", [(2, 21, 20), (1, 8, 16), (3, 37, 53), (1, 200, 176)])
def test_weight_gradient_straight_from_nhwc(ksize, shape):
    """transpose-read kernel: dW from the NHWC tensors themselves, vs torch's float64 weight gradient; ragged tiles; repeatable"""
    L, lib = _lib()
    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(7 + ksize + H)
    x = _bf16_nhwc(torch.randn(B, 128, H, W, device="cuda", generator=g))
    dy = _bf16_nhwc(torch.randn(B, 128, H, W, device="cuda", generator=g))
    ws = torch.empty(int(lib.v3d_dense_train_wgrad_workspace(ksize)), dtype=torch.uint8, device="cuda")
    dw = torch.empty((128, 128, ksize, ksize), device="cuda")
    L.check(lib.v3d_dense_train_wgrad(L.ptr(x), L.ptr(dy), B, H, W, ksize, L.ptr(dw), L.ptr(ws), ws.numel(), L.stream_ptr()), "wgrad")
    wref = torch.zeros((128, 128, ksize, ksize), device="cuda", dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wref, padding=ksize // 2).backward(dy.double())
    assert _rel(dw, wref.grad) < 1e-5, _rel(dw, wref.grad)  # fp32 accumulation of exact bf16 products
    dw2 = torch.empty_like(dw)
    L.check(lib.v3d_dense_train_wgrad(L.ptr(x), L.ptr(dy), B, H, W, ksize, L.ptr(dw2), L.ptr(ws), ws.numel(), L.stream_ptr()), "wgrad")
    assert torch.equal(dw, dw2)


def test_batchnorm_relu_forward_backward():
    L, lib = _lib()
    B, H, W = 2, 18, 12
    M = B * H * W
    g = torch.Generator(device="cuda").manual_seed(9)
    x = _bf16_nhwc(torch.randn(B, 128, H, W, device="cuda", generator=g) * 1.5 + 0.3)
    dy = _bf16_nhwc(torch.randn(B, 128, H, W, device="cuda", generator=g))
    gamma = torch.rand(128, device="cuda", generator=g) + 0.5
    beta = torch.randn(128, device="cuda", generator=g) * 0.2
    eps, mom = 1e-3, 0.01
    # statistics from per-tile partials (as the convolution's epilogue writes them)
    xf = x.float().permute(0, 2, 3, 1).reshape(M, 128)
    tiles = (M + 127) // 128
    part = torch.zeros((tiles, 2, 128), device="cuda")
    for t in range(tiles):
        blk = xf[t * 128:(t + 1) * 128]
        part[t, 0], part[t, 1] = blk.sum(0), (blk * blk).sum(0)
    mean, invstd = torch.empty(128, device="cuda"), torch.empty(128, device="cuda")
    rm, rv = torch.zeros(128, device="cuda"), torch.ones(128, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    L.check(lib.v3d_dense_train_bn_finalize(L.ptr(part), tiles, M, eps, mom, L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv), L.ptr(nbt),
                                            L.stream_ptr()), "bn_finalize")
    bn = torch.nn.BatchNorm2d(128, eps=eps, momentum=mom).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xr = x.float().requires_grad_(True)
    yref = torch.relu(bn(xr))
    assert torch.allclose(mean, xf.mean(0), atol=1e-5) and torch.allclose(invstd, torch.rsqrt(xf.var(0, unbiased=False) + eps), rtol=1e-5)
    assert torch.allclose(rm, bn.running_mean, atol=1e-6) and torch.allclose(rv, bn.running_var, rtol=1e-5) and int(nbt) == 1
    y = torch.empty_like(x)
    L.check(lib.v3d_dense_train_bn_relu_apply(L.ptr(x), M, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), 1, L.ptr(y), L.stream_ptr()),
            "bn_apply")
    assert _rel(y.float(), yref) < BF16_EPS
    yref.backward(dy.float())
    dx, dgam, dbet = torch.empty_like(x), torch.empty(128, device="cuda"), torch.empty(128, device="cuda")
    ws = torch.empty(int(lib.v3d_dense_train_bn_bwd_workspace()), dtype=torch.uint8, device="cuda")
    L.check(lib.v3d_dense_train_bn_relu_bwd(L.ptr(x), L.ptr(dy), M, L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(beta), 1, L.ptr(dx),
                                            L.ptr(dgam), L.ptr(dbet), L.ptr(ws), ws.numel(), L.stream_ptr()), "bn_bwd")
    assert _rel(dgam, bn.weight.grad) < 1e-4 and _rel(dbet, bn.bias.grad) < 1e-4
    assert _rel(dx.float(), xr.grad) < 2 * BF16_EPS


@pytest.mark.parametrize("O", [16, 48])
def test_head_forward_backward(O):
    L, lib = _lib()
    B, H, W = 2, 15, 12
    g = torch.Generator(device="cuda").manual_seed(O)
    feat = _bf16_nhwc(torch.randn(B, 128, H, W, device="cuda", generator=g))
    w = (torch.randn(O, 128, device="cuda", generator=g) * 0.05).contiguous()
    b = torch.randn(O, device="cuda", generator=g) * 0.1
    maps = torch.empty((B, O, H, W), device="cuda")
    L.check(lib.v3d_dense_train_head_fwd(L.ptr(feat), B, H, W, L.ptr(w), L.ptr(b), O, L.ptr(maps), L.stream_ptr()), "head_fwd")
    fr = feat.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(fr, wr.view(O, 128, 1, 1), br)
    assert _rel(maps, ref) < 1e-5
    dm = torch.randn(B, O, H, W, device="cuda", generator=g)
    ref.backward(dm)
    dfeat, dw, db = torch.empty_like(feat), torch.empty_like(w), torch.empty_like(b)
    ws = torch.empty(int(lib.v3d_dense_train_head_workspace(O)), dtype=torch.uint8, device="cuda")
    L.check(lib.v3d_dense_train_head_bwd(L.ptr(feat), L.ptr(dm), B, H, W, L.ptr(w), O, L.ptr(dfeat), L.ptr(dw), L.ptr(db), L.ptr(ws),
                                         ws.numel(), L.stream_ptr()), "head_bwd")
    assert _rel(dw, wr.grad) < 1e-5 and _rel(db, br.grad) < 1e-5
    assert _rel(dfeat.float(), fr.grad) < BF16_EPS


def _stack(seed, B, H, W):
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector.proposal import ProposalLayer
    from vision3d_amd.detector.second import RPN
    torch.manual_seed(seed)
    rpn, head = RPN().cuda().train(), ProposalLayer(second_car_cfg()).cuda().train()
    with torch.no_grad():
        for m in rpn.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.6, 1.4)
                m.bias.uniform_(-0.2, 0.2)
        head.conv_cls.weight.normal_(std=0.05)
        head.conv_reg.weight.normal_(std=0.05)
    bev = torch.randn(B, 128, H, W, device="cuda")
    bev = bev * (torch.rand(B, 1, H, W, device="cuda") < 0.3)  # sparse like a BEV map
    return rpn, head, bev


def test_whole_dense_stack_matches_the_torch_modules_within_bf16_tolerance():
    """7 x (conv + BatchNorm(batch stats) + ReLU) + heads: maps, input gradient, all 25 parameter gradients and the running
    statistics vs the fp32 torch modules; bit-repeatable; a stale backward raises."""
    import copy
    from vision3d_amd import dense_train
    B, H, W = 2, 40, 32
    rpn, head, bev = _stack(0, B, H, W)
    rpn_ref, head_ref = copy.deepcopy(rpn), copy.deepcopy(head)
    x = _bf16_nhwc(bev).requires_grad_(True)
    assert dense_train.supported(rpn, head, x)
    cache = {}
    maps = dense_train.train_head_maps(rpn, head, x, cache)
    gmap = torch.randn_like(maps)
    (maps * gmap).sum().backward()
    # fp32 reference on the same bf16-rounded input, and torch's OWN bf16 autocast run of the same modules as the yardstick:
    # the native path must be as close to fp32 as autocast is (relative L2 error; a max-norm figure is dominated by the few
    # ReLU decisions that flip when a pre-activation near zero is rounded)
    def l2(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / b.norm().clamp_min(1e-30))

    def torch_run(autocast):
        r, h = copy.deepcopy(rpn_ref), copy.deepcopy(head_ref)
        xin = x.detach().float().contiguous().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            f = r.up_block(r.down_block(xin))
            out = torch.cat((h.conv_cls(f), h.conv_reg(f)), 1)
        (out.float() * gmap).sum().backward()
        grads = {n: p.grad for n, p in list(r.named_parameters()) + list(h.named_parameters())}
        return out.float(), xin.grad, grads, r

    ref, xg_ref, g_ref, rpn_done = torch_run(False)
    ac, xg_ac, g_ac, _ = torch_run(True)
    e_maps, e_x = l2(maps, ref), l2(x.grad.float(), xg_ref)
    assert e_maps < max(0.02, 2 * l2(ac, ref)), (e_maps, l2(ac, ref))
    assert e_x < max(0.04, 2 * l2(xg_ac, xg_ref)), (e_x, l2(xg_ac, xg_ref))
    named = dict(list(rpn.named_parameters()) + list(head.named_parameters()))
    assert len(named) == 7 * 3 + 4
    worst = 0.0
    for n, p in named.items():
        assert p.grad is not None, n
        e, e_ac = l2(p.grad, g_ref[n]), l2(g_ac[n], g_ref[n])
        worst = max(worst, e)
        assert e < max(0.04, 2 * e_ac), (n, e, e_ac)
    print("dense train stack vs fp32: maps %.3e (autocast %.3e), d(input) %.3e (autocast %.3e), worst parameter gradient %.3e"
          % (e_maps, l2(ac, ref), e_x, l2(xg_ac, xg_ref), worst))
    rpn_ref = rpn_done
    for (n, b), (_, br) in zip(rpn.named_buffers(), rpn_ref.named_buffers()):
        if "num_batches" in n:
            assert int(b) == int(br) == 1
        else:  # running mean / variance: same momentum update from (nearly) the same batch statistics
            assert b._version > 0 and _rel(b, br) < 0.02, (n, _rel(b, br))
    # repeatability: same input, fresh parameter copies -> identical maps and gradients
    rpn2, head2, _ = _stack(0, B, H, W)
    x2 = x.detach().clone().requires_grad_(True)
    maps2 = dense_train.train_head_maps(rpn2, head2, x2, {})
    (maps2 * gmap).sum().backward()
    assert torch.equal(maps, maps2) and torch.equal(x.grad, x2.grad)
    for (n, p), (_, p2) in zip(list(rpn.named_parameters()), list(rpn2.named_parameters())):
        assert torch.equal(p.grad, p2.grad), n
    # generation guard
    m1 = dense_train.train_head_maps(rpn, head, x, cache)
    m2 = dense_train.train_head_maps(rpn, head, x, cache)
    with pytest.raises(RuntimeError, match="belongs to forward"):
        m1.sum().backward()
    m2.sum().backward()


def test_second_train_step_takes_the_native_dense_path_and_records_no_miopen_convolution():
    """Second.forward in training mode under bf16 autocast: sparse plan -> native dense plan; the profiler sees the dt_* kernels
    and no MIOpen / igemm / batch-norm kernel of torch."""
    from vision3d_amd import synth
    from vision3d_amd.core import Preprocessor, ProposalTargetAssigner
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import ProposalLoss, Second
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = Second(cfg).cuda().train()
    pre, assigner, loss_fn = Preprocessor(cfg, seed=0), ProposalTargetAssigner(cfg), ProposalLoss(cfg)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (0, 1)]
    tg = []
    for s in (0, 1):
        gt = torch.from_numpy(synth.make_gt_boxes(s))
        tg.append(assigner(dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long), box_ignore=torch.zeros(len(gt), dtype=torch.bool))))
    tgt = {k: torch.stack([t[k] for t in tg]).cuda() for k in ("G_cls", "G_reg", "M_cls", "M_reg")}

    def step():
        item = pre(dict(points=clouds))
        item.update(tgt)
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses = loss_fn(model(item))
        losses["loss"].backward()
        return losses["loss"].detach()

    l0 = step()
    assert "_dense_train_plans" in model.__dict__ and len(model.__dict__["_dense_train_plans"]) == 1
    assert torch.isfinite(l0)
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    names = {e.key for e in prof.key_averages()}
    assert any(n.startswith("void dt_conv_kernel") or "dt_conv_kernel" in n for n in names), sorted(names)[:40]
    bad = [n for n in names if any(t in n.lower() for t in ("igemm", "miopen", "naive_conv", "batch_norm", "batchnorm")) and "dt_" not in n]
    assert not bad, bad


# ------------------------------------------------------------------------------------------------ the fp32-class ("bf16x3") step
@pytest.mark.parametrize("smooth", [True, False], ids=["relu_inactive", "relu_active"])
def test_whole_dense_stack_fp32_class_matches_the_fp32_torch_modules(smooth):
    """dense_train precision "bf16x3" (split hi + lo storage, three-term products): maps, input gradient, all 25 parameter gradients
    and the running statistics against the fp32 torch modules of the same stack.
      relu_inactive  BatchNorm offsets of +8 keep every pre-activation positive: the stack is smooth, and every error is the
                     arithmetic's own -- relative L2 of 1e-5 .. 1e-4 (2^-17 per product through 7 normalised layers; a missing
                     hi * lo term anywhere would show as 2e-3);
      relu_active    the usual offsets: two fp32-class computations disagree on the sign of the few pre-activations that lie within
                     their rounding of zero, and each such ReLU decision moves gradient entries by their whole value -- a relative
                     L2 error of sqrt(fraction flipped), a few 1e-3, in the gradients (the forward maps stay at 1e-5).
    Bit-repeatable; a stale backward raises."""
    import copy
    from vision3d_amd import dense_train
    B, H, W = 2, 40, 32
    rpn, head, bev = _stack(0, B, H, W)
    if smooth:
        with torch.no_grad():
            for m in rpn.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.bias.add_(8.0)
    rpn_ref, head_ref = copy.deepcopy(rpn), copy.deepcopy(head)
    x = bev.clone().requires_grad_(True)
    assert dense_train.supported(rpn, head, x, "bf16x3")
    cache = {}
    maps = dense_train.train_head_maps(rpn, head, x, cache, "bf16x3")
    gmap = torch.randn_like(maps)
    (maps * gmap).sum().backward()

    def l2(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / b.norm().clamp_min(1e-30))

    xin = bev.clone().requires_grad_(True)
    f = rpn_ref.up_block(rpn_ref.down_block(xin))
    ref = torch.cat((head_ref.conv_cls(f), head_ref.conv_reg(f)), 1)
    (ref * gmap).sum().backward()
    g_ref = {n: p.grad for n, p in list(rpn_ref.named_parameters()) + list(head_ref.named_parameters())}
    e_maps, e_x = l2(maps, ref), l2(x.grad, xin.grad)
    named = dict(list(rpn.named_parameters()) + list(head.named_parameters()))
    assert len(named) == 7 * 3 + 4
    errs = {n: l2(p.grad, g_ref[n]) for n, p in named.items()}
    worst = max(errs.values())
    print("dense train stack (%s), bf16x3 vs fp32 torch modules: maps %.3e, d(input) %.3e, worst parameter gradient %.3e (%s)"
          % ("relu inactive" if smooth else "relu active", e_maps, e_x, worst, max(errs, key=errs.get)))
    assert e_maps < 1e-4, e_maps
    if not smooth:
        assert e_x < 2e-2 and worst < 2e-2, (e_x, errs)
    else:
        # The yardstick for the parameter gradients is the fp32 modules' OWN error against a float64 run of the same stack: some of
        # them are badly conditioned (a BatchNorm offset feeds a convolution whose output the next BatchNorm re-centres: its gradient
        # is a sum that cancels to ~1e-3 of its terms), and a 2^-17 arithmetic may be 2^7 times further out than a 2^-24 one.
        r64, h64 = copy.deepcopy(rpn_ref).double(), copy.deepcopy(head_ref).double()
        for m in list(r64.parameters()) + list(h64.parameters()):
            m.grad = None
        x64 = bev.double().requires_grad_(True)
        f64 = r64.up_block(r64.down_block(x64))
        (torch.cat((h64.conv_cls(f64), h64.conv_reg(f64)), 1) * gmap.double()).sum().backward()
        g64 = {n: p.grad for n, p in list(r64.named_parameters()) + list(h64.named_parameters())}
        assert l2(x.grad, x64.grad) < 2e-4, l2(x.grad, x64.grad)
        for n, p in named.items():
            mine, torch32 = l2(p.grad, g64[n]), l2(g_ref[n], g64[n])
            assert mine < max(2e-4, 256 * torch32), (n, mine, torch32)
    for (n, b), (_, br) in zip(rpn.named_buffers(), rpn_ref.named_buffers()):
        if "num_batches" in n:
            assert int(b) == int(br) == 1
        else:
            assert b._version > 0 and _rel(b, br) < 1e-5, (n, _rel(b, br))
    # repeatability: same input, fresh parameter copies -> identical maps and gradients
    rpn2, head2, _ = _stack(0, B, H, W)
    if smooth:
        with torch.no_grad():
            for m in rpn2.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.bias.add_(8.0)
    x2 = bev.clone().requires_grad_(True)
    maps2 = dense_train.train_head_maps(rpn2, head2, x2, {}, "bf16x3")
    (maps2 * gmap).sum().backward()
    assert torch.equal(maps, maps2) and torch.equal(x.grad, x2.grad)
    for (n, p), (_, p2) in zip(list(rpn.named_parameters()), list(rpn2.named_parameters())):
        assert torch.equal(p.grad, p2.grad), n
    m1 = dense_train.train_head_maps(rpn, head, x, cache, "bf16x3")
    m2 = dense_train.train_head_maps(rpn, head, x, cache, "bf16x3")
    with pytest.raises(RuntimeError, match="belongs to forward"):
        m1.sum().backward()
    m2.sum().backward()


def test_reference_fp32_train_step_runs_native_and_records_no_miopen_convolution():
    """The reference's train.py:58-66 as written -- fp32, no autocast, nothing opted into: Second.forward in training mode takes the
    sparse training plan and the fp32-class dense plan; the profiler sees the hand-written kernels and no MIOpen / igemm /
    batch-norm kernel of torch; loss and gradients equal those of the torch modules (dense_train_precision = "torch") to fp32-class
    accuracy."""
    from vision3d_amd import synth
    from vision3d_amd.core import Preprocessor, ProposalTargetAssigner
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import ProposalLoss, Second
    cfg = second_car_cfg()
    pre, assigner, loss_fn = Preprocessor(cfg, seed=0), ProposalTargetAssigner(cfg), ProposalLoss(cfg)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (0, 1)]
    tg = []
    for s in (0, 1):
        gt = torch.from_numpy(synth.make_gt_boxes(s))
        tg.append(assigner(dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long), box_ignore=torch.zeros(len(gt), dtype=torch.bool))))
    tgt = {k: torch.stack([t[k] for t in tg]).cuda() for k in ("G_cls", "G_reg", "M_cls", "M_reg")}

    def make(precision=None):
        torch.manual_seed(0)
        model = Second(cfg).cuda().train()
        if precision is not None:
            model.dense_train_precision = precision
        return model

    def step(model):
        item = pre(dict(points=clouds))
        item.update(tgt)
        model.zero_grad(set_to_none=True)
        out = model(item)  # train.py:63 -- no autocast anywhere
        losses = loss_fn(out)
        losses["loss"].backward()
        return losses["loss"].detach(), "_head_maps" in out

    model = make()
    assert model.dense_train_precision == "bf16x3"
    l0, fused = step(model)
    assert fused and model.torch_dense_fallbacks == 0 and torch.isfinite(l0)
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step(model)
        torch.cuda.synchronize()
    names = {e.key for e in prof.key_averages()}
    assert any("dt_wgrad_kernel" in n for n in names) and any("conv2d_bf16x3" in n for n in names), sorted(names)[:60]
    bad = [n for n in names if any(t in n.lower() for t in ("igemm", "miopen", "naive_conv", "batch_norm", "batchnorm")) and "dt_" not in n]
    assert not bad, bad
    ref = make("torch")
    l1, fused1 = step(ref)
    assert not fused1
    assert abs(float(l0) - float(l1)) <= 1e-5 * abs(float(l1)), (float(l0), float(l1))
    worst = 0.0
    for n, p in ref.named_parameters():
        e = float((grads[n].double() - p.grad.double()).norm() / p.grad.double().norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 2e-2, (n, e)  # (ReLU decisions within rounding of zero: see the stack test)
    print("fp32 train step, native bf16x3 dense half vs torch modules: loss %.6f vs %.6f, worst parameter gradient rel L2 %.3e" % (float(l0), float(l1), worst))
