"""GPU: the fused proposal loss (csrc/proposal_loss.hip) against the torch expressions of ProposalLoss (vision3d/detector/proposal.py:
100-141, restated in vision3d_amd/detector/proposal.py and pinned to reference outputs in tests/test_host_golden.py): loss terms and the
gradient with respect to the fused head maps, degenerate inputs, upstream gradients, repeatability."""
import pytest
import torch

from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


def _inputs(seed, b=2, h=20, w=24, n_cls=1, n_yaw=2, positives=True, big=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    o = n_cls * n_yaw * 8
    maps = torch.randn(b, o, h, w, device="cuda", generator=g) * (12.0 if big else 1.5)
    shape = (b, n_cls, n_yaw, h, w)
    r = torch.rand(shape, device="cuda", generator=g)
    G_cls = (r > 0.9).to(torch.int8) if positives else torch.zeros(shape, dtype=torch.int8, device="cuda")
    M_cls = torch.rand(shape, device="cuda", generator=g) > 0.2
    G_reg = torch.randn(shape + (7,), device="cuda", generator=g) * 1.2
    M_reg = (G_cls == 1).unsqueeze(-1)
    return maps, dict(G_cls=G_cls, M_cls=M_cls, G_reg=G_reg, M_reg=M_reg)


def _torch_loss(cfg, maps, tg):
    from vision3d_amd.detector import ProposalLoss
    b, o, h, w = maps.shape
    na = cfg.NUM_CLASSES * cfg.NUM_YAW
    P_cls = maps[:, :na].reshape(b, cfg.NUM_CLASSES, cfg.NUM_YAW, h, w)
    P_reg = maps[:, na:].reshape(b, cfg.NUM_CLASSES, 7, cfg.NUM_YAW, h, w).permute(0, 1, 3, 4, 5, 2)
    item = dict(tg, P_cls=P_cls, P_reg=P_reg)
    return ProposalLoss(cfg)(item)


def _fused_loss(cfg, maps, tg):
    from vision3d_amd.detector import ProposalLoss
    item = dict(tg, _head_maps=maps, P_cls=None, P_reg=None)
    return ProposalLoss(cfg)(item)


@pytest.mark.parametrize("case", ["plain", "no_positives", "large_logits", "int64_targets"])
def test_fused_loss_and_gradient_match_the_torch_expressions(case):
    cfg = second_car_cfg()
    maps, tg = _inputs(3, positives=case != "no_positives", big=case == "large_logits")
    if case == "int64_targets":
        tg["G_cls"] = tg["G_cls"].long()
    ref_in = maps.clone().double().requires_grad_(True)
    ref = _torch_loss(cfg, ref_in, {k: (v.double() if v.dtype == torch.float32 else v) for k, v in tg.items()})
    (2.0 * ref["cls_loss"] + 3.0 * ref["reg_loss"]).backward()
    got_in = maps.clone().requires_grad_(True)
    got = _fused_loss(cfg, got_in, tg)
    assert isinstance(got["loss"].grad_fn, object) and got["cls_loss"].shape == ()
    (2.0 * got["cls_loss"] + 3.0 * got["reg_loss"]).backward()
    for k in ("cls_loss", "reg_loss", "loss"):
        assert abs(float(got[k].detach()) - float(ref[k].detach())) <= 2e-6 * max(1.0, abs(float(ref[k].detach()))), k
    gr = ref_in.grad.float()
    assert float((got_in.grad - gr).abs().max()) <= 1e-6 * max(float(gr.abs().max()), 1e-30) + 1e-12
    if case == "no_positives":
        assert float(got["reg_loss"]) == 0.0 and float(got_in.grad[:, cfg.NUM_CLASSES * cfg.NUM_YAW:].abs().max()) == 0.0


def test_fused_loss_is_bit_repeatable_and_matches_fp32_torch_on_a_full_map():
    cfg = second_car_cfg()
    maps, tg = _inputs(11, b=4, h=200, w=176)
    outs = []
    for _ in range(2):
        x = maps.clone().requires_grad_(True)
        l = _fused_loss(cfg, x, tg)
        l["loss"].backward()
        outs.append((l["loss"].detach().clone(), x.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    x = maps.clone().requires_grad_(True)
    ref = _torch_loss(cfg, x, tg)
    ref["loss"].backward()
    assert abs(float(outs[0][0]) - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    assert float((outs[0][1] - x.grad).abs().max()) <= 1e-5 * float(x.grad.abs().max())


def test_second_train_forward_hands_its_fused_maps_to_the_loss():
    """Second.forward on the native training path leaves `_head_maps` in the item; ProposalLoss then runs the native pass and the
    parameter gradients equal those of the torch loss on the same step (same dense / sparse kernels underneath)."""
    from vision3d_amd import synth
    from vision3d_amd.core import Preprocessor, ProposalTargetAssigner
    from vision3d_amd.detector import ProposalLoss, Second
    cfg = second_car_cfg()
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in range(2)]
    assigner = ProposalTargetAssigner(cfg)
    targets = []
    for s in range(2):
        gt = torch.from_numpy(synth.make_gt_boxes(s))
        targets.append(assigner(dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long), box_ignore=torch.zeros(len(gt), dtype=torch.bool))))
    tgt = {k: torch.stack([t[k] for t in targets]).cuda() for k in ("G_cls", "G_reg", "M_cls", "M_reg")}

    def run(fused):
        torch.manual_seed(0)
        model = Second(cfg).cuda().train()
        item = Preprocessor(cfg, seed=0)(dict(points=[c.clone() for c in clouds]))
        item.update(tgt)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(item)
            assert "_head_maps" in out
            if not fused:
                out = {k: v for k, v in out.items() if k != "_head_maps"}
            losses = ProposalLoss(cfg)(out)
        losses["loss"].backward()
        return float(losses["loss"]), {n: p.grad.clone() for n, p in model.named_parameters()}
    la, ga = run(True)
    lb, gb = run(False)
    assert abs(la - lb) <= 1e-5 * abs(lb)
    for n in ga:  # the two head-map gradients differ by fp32 rounding; behind the bf16 dense backward that becomes bf16 ulp flips
        rel = float((ga[n] - gb[n]).norm() / gb[n].norm().clamp_min(1e-30))
        assert rel <= 1e-2, (n, rel)


def test_reference_fp32_script_reaches_the_native_dense_kernels_when_opted_in():
    """The reference's train.py:58-66 runs fp32 with no autocast.  With `model.dense_train_precision = "bf16"` the training forward
    enters bf16 autocast itself for the dense half: the step takes the native kernels (fused maps present, no torch fallback) and
    loss / gradients equal those of the same step wrapped in autocast by the caller.  "torch" keeps the torch modules and hands no
    fused maps over (the default, "bf16x3", is the native fp32-class step: tests/test_gpu_dense_train.py); a re-used item never carries maps of an earlier forward."""
    from vision3d_amd import synth
    from vision3d_amd.core import Preprocessor, ProposalTargetAssigner
    from vision3d_amd.detector import ProposalLoss, Second
    cfg = second_car_cfg()
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in range(2)]
    assigner = ProposalTargetAssigner(cfg)
    targets = []
    for s in range(2):
        gt = torch.from_numpy(synth.make_gt_boxes(s))
        targets.append(assigner(dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long), box_ignore=torch.zeros(len(gt), dtype=torch.bool))))
    tgt = {k: torch.stack([t[k] for t in targets]).cuda() for k in ("G_cls", "G_reg", "M_cls", "M_reg")}

    def run(mode):
        torch.manual_seed(0)
        model = Second(cfg).cuda().train()
        item = Preprocessor(cfg, seed=0)(dict(points=[c.clone() for c in clouds]))
        item.update(tgt)
        if mode == "caller_autocast":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(item)
                losses = ProposalLoss(cfg)(out)
        else:
            model.dense_train_precision = "bf16" if mode == "opt_in" else "torch"
            out = model(item)  # train.py:63 -- no autocast anywhere
            losses = ProposalLoss(cfg)(out)
        fused = "_head_maps" in out
        losses["loss"].backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters()}
        # the same dict through the eval path: the training maps must be gone
        model.eval()
        with torch.no_grad():
            out2 = model(out)
        assert "_head_maps" not in out2
        return float(losses["loss"]), grads, fused, model.torch_dense_fallbacks

    la, ga, fa, fb_a = run("caller_autocast")
    lb, gb, fb, fb_b = run("opt_in")
    lc, _, fc, _ = run("default")
    assert fa and fb and not fc and fb_a == 0 and fb_b == 0
    assert la == lb
    for n in ga:
        assert torch.equal(ga[n], gb[n]), n
    assert abs(lc - la) <= 5e-2 * abs(lc)  # fp32 torch modules vs bf16 storage: same step, different precision contract
