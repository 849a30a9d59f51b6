"""Training plan (csrc/second_plan.hip "Training plan", runtime.PlanTrainFunction): the sparse half of a train step as one
native call each way must agree with the module-by-module autograd path (spconv.SparseSequential: SparseConvFunction +
SparseBatchNormReLUFunction + .dense()), which test_gpu_spconv.py / test_gpu_configs.py pin against torch and fp64."""
import copy

import numpy as np
import pytest
import torch

from vision3d_amd import synth
from vision3d_amd.core import Preprocessor
from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


def _model_and_item(bs, seed=0):
    from vision3d_amd.detector import Second
    cfg = second_car_cfg()
    torch.manual_seed(seed)
    model = Second(cfg).cuda().train()
    with torch.no_grad():  # non-trivial BatchNorm parameters and statistics
        for m in model.cnn.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.5, 2.0)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in range(bs)]
    item = Preprocessor(cfg, seed=0)(dict(points=clouds))
    return cfg, model, item


def _run(cnn, item, g_bev, native):
    cnn.native_train = native
    for p in cnn.parameters():
        p.grad = None
    bev = cnn(item["voxel_mean"], item["coordinates"], item["batch_size"])
    (bev * g_bev).sum().backward()
    grads = {n: p.grad.detach().clone() for n, p in cnn.named_parameters()}
    stats = {n: b.detach().clone() for n, b in cnn.named_buffers() if "running" in n or "num_batches" in n}
    return bev.detach().clone(), grads, stats


@pytest.mark.parametrize("bs", [1, 2])
def test_train_plan_matches_module_path(bs):
    cfg, model, item = _model_and_item(bs)
    cnn_a, cnn_b = model.cnn, copy.deepcopy(model.cnn)
    torch.manual_seed(1)
    g_bev = torch.randn(bs, 128, 200, 176, device="cuda")
    bev_a, grads_a, stats_a = _run(cnn_a, item, g_bev, native=True)
    bev_b, grads_b, stats_b = _run(cnn_b, item, g_bev, native=False)
    assert "_train_plans" in cnn_a.__dict__ and "_train_plans" not in cnn_b.__dict__  # the two paths really differ
    scale = float(bev_b.abs().max())
    assert scale > 0
    assert float((bev_a - bev_b).abs().max()) <= 1e-5 * scale
    assert abs(int((bev_a != 0).sum()) - int((bev_b != 0).sum())) <= 8  # values at the ReLU boundary may fall either side
    assert set(grads_a) == set(grads_b) and len(grads_a) == 14 * 3
    for n in grads_b:  # same kernels on both paths (the plan tunes from a coordinate-only pass before its first step)
        ref = grads_b[n]
        err = float((grads_a[n] - ref).abs().max())
        assert err <= 1e-5 * float(ref.abs().max()) + 1e-12, (n, err, float(ref.abs().max()))
    for n in stats_b:  # running statistics: momentum update with the unbiased variance, num_batches_tracked += 1
        if "num_batches" in n:
            assert int(stats_a[n]) == int(stats_b[n]) == 1, n
        else:
            torch.testing.assert_close(stats_a[n], stats_b[n], rtol=1e-5, atol=1e-6, msg=n)


def test_train_plan_repeats_bit_for_bit_and_tracks_parameter_updates():
    cfg, model, item = _model_and_item(2)
    cnn = model.cnn
    torch.manual_seed(1)
    g_bev = torch.randn(2, 128, 200, 176, device="cuda")
    snapshot = copy.deepcopy(cnn.state_dict())
    bev_1, grads_1, _ = _run(cnn, item, g_bev, native=True)  # first call: tunes the kernel choice from a coordinate-only pass
    cnn.load_state_dict(snapshot)
    bev_2, grads_2, _ = _run(cnn, item, g_bev, native=True)
    cnn.load_state_dict(snapshot)
    bev_3, grads_3, _ = _run(cnn, item, g_bev, native=True)
    assert torch.equal(bev_2, bev_3)  # deterministic: no atomics in any reduction
    for n in grads_2:
        assert torch.equal(grads_2[n], grads_3[n]), n
    assert torch.equal(bev_1, bev_2)  # ... so even the first step runs the kernels of every later step
    with torch.no_grad():  # an optimiser step: the plan reads the parameters on every call
        for p in cnn.parameters():
            p.mul_(0.5)
    bev_4, _, _ = _run(cnn, item, g_bev, native=True)
    assert not torch.equal(bev_4, bev_3)
    ref = copy.deepcopy(cnn)
    ref.load_state_dict(cnn.state_dict())
    # (same parameters, module path, running statistics irrelevant for training-mode outputs)
    bev_5, _, _ = _run(ref, item, g_bev, native=False)
    assert float((bev_4 - bev_5).abs().max()) <= 1e-5 * float(bev_5.abs().max())


def test_train_plan_step_is_capturable_in_a_hip_graph():
    """No host read anywhere in the plan's forward / backward: one train step of the sparse half replays from a graph."""
    cfg, model, item = _model_and_item(1)
    cnn = model.cnn
    torch.manual_seed(1)
    g_bev = torch.randn(1, 128, 200, 176, device="cuda")
    plan = cnn._train_plan(item["voxel_mean"].shape[0], 1, item["voxel_mean"].device)
    bev_eager = plan.train_forward(item["voxel_mean"], item["coordinates"], 1)
    grads_eager = [g.clone() for g in plan.train_backward(g_bev, 1)]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        bev = plan.train_forward(item["voxel_mean"], item["coordinates"], 1)
        grads = plan.train_backward(g_bev, 1)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(bev, bev_eager)
    for a, b in zip(grads, grads_eager):
        assert torch.equal(a, b)


def test_train_plan_bf16_channels_last_bev_is_the_cast_of_the_fp32_map():
    """Under bf16 autocast the plan hands the RPN a bfloat16 channels_last BEV map and takes the gradient in that form:
    same numbers as casting / re-laying-out the float32 NCHW map in torch."""
    cfg, model, item = _model_and_item(2)
    cnn = model.cnn
    vm, co = item["voxel_mean"], item["coordinates"]
    plan = cnn._train_plan(vm.shape[0], 2, vm.device)
    bev32 = plan.train_forward(vm, co, 2)
    bev16 = plan.train_forward(vm, co, 2, bf16_nhwc=True)
    assert bev16.dtype == torch.bfloat16 and bev16.shape == bev32.shape
    assert bev16.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(bev16, bev32.to(torch.bfloat16))
    torch.manual_seed(2)
    g16 = torch.randn(2, 128, 200, 176, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    grads16 = [g.clone() for g in plan.train_backward(g16, 2)]
    grads32 = plan.train_backward(g16.float().contiguous(), 2)
    for a, b in zip(grads16, grads32):
        assert torch.equal(a, b)
    # and through the module entry point: autocast picks the bf16 form, the RPN's first convolution takes it as is
    with torch.autocast("cuda", dtype=torch.bfloat16):
        bev = cnn(vm, co, 2)
        feat = model.rpn(bev)
    assert bev.dtype == torch.bfloat16 and bev.is_contiguous(memory_format=torch.channels_last)
    feat.float().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in cnn.parameters())


def test_rpn_training_forward_equals_padded_module_sequence():
    """RPN.forward in training fuses ZeroPad2d(1) into the first convolution: same values and gradients as the module list."""
    from vision3d_amd.detector.second import RPN
    torch.manual_seed(0)
    rpn = RPN().cuda().train()
    x = torch.randn(2, 128, 40, 36, device="cuda", requires_grad=True)
    y = rpn(x)
    (gx,) = torch.autograd.grad(y.square().sum(), x)
    rpn2 = copy.deepcopy(rpn)
    rpn2.load_state_dict(rpn.state_dict())
    for m in rpn2.modules():  # undo the running-statistics update of the first call
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = rpn2.up_block(rpn2.down_block(x2))
    (gx2,) = torch.autograd.grad(y2.square().sum(), x2)
    torch.testing.assert_close(y, y2, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gx, gx2, rtol=1e-3, atol=1e-3 * float(gx2.abs().max()))


def test_dense_gradients_are_complete_when_the_plan_backward_starts():
    """dist_util.TwoPhaseGradReducer reduces the dense half's bucket from `plan.pre_backward_hook`, i.e. while the native sparse
    backward runs: every RPN / head gradient must already hold its final value at that moment (under bf16 autocast too)."""
    from vision3d_amd.core import ProposalTargetAssigner
    from vision3d_amd.detector import ProposalLoss
    cfg, model, item = _model_and_item(2)
    assigner, targets = ProposalTargetAssigner(cfg), []
    for f in range(2):
        gt = torch.from_numpy(synth.make_gt_boxes(f))
        targets.append(assigner(dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long),
                                     box_ignore=torch.zeros(len(gt), dtype=torch.bool))))
    item.update({k: torch.stack([t[k] for t in targets]).cuda() for k in ("G_cls", "G_reg", "M_cls", "M_reg")})
    sparse_ids = {id(p) for p in model.cnn.parameters()}
    dense = [p for p in model.parameters() if id(p) not in sparse_ids]
    snap = []
    with torch.autocast("cuda", dtype=torch.bfloat16):
        losses = ProposalLoss(cfg)(model(item))
    plans = list(model.cnn.__dict__["_train_plans"].values())
    assert len(plans) == 1
    plans[0].pre_backward_hook = lambda: snap.append([None if p.grad is None else p.grad.detach().clone() for p in dense])
    losses["loss"].backward()
    assert len(snap) == 1 and len(dense) == 25
    for at_hook, p in zip(snap[0], dense):
        assert at_hook is not None and torch.equal(at_hook, p.grad)
    assert all(p.grad is not None for p in model.cnn.parameters())


def test_plan_state_belongs_to_one_forward_and_the_capacity_word_is_read_every_step():
    """(1) the plan holds the saved activations / rulebooks of ONE forward -- a backward that belongs to an
    earlier forward (two train-mode forwards before a backward) raises instead of returning the gradients of the wrong
    frame; (2) the capacity-overflow word of EVERY step is checked (copied to pinned memory behind the
    step, read when the next step begins), not only the first one; (3) a train forward bumps the version counters of the
    running statistics it updated through raw pointers, so caches keyed on (data_ptr, _version) see it."""
    cfg, model, item = _model_and_item(1)
    cnn = model.cnn
    bn = next(m for m in cnn.modules() if isinstance(m, torch.nn.BatchNorm1d))
    v0 = bn.running_mean._version
    bev1 = cnn(item["voxel_mean"], item["coordinates"], item["batch_size"])
    assert bn.running_mean._version > v0 and bn.num_batches_tracked._version > 0
    bev2 = cnn(item["voxel_mean"], item["coordinates"], item["batch_size"])   # overwrites the plan's saved state
    with pytest.raises(RuntimeError, match="belongs to forward"):
        bev1.sum().backward()
    bev2.sum().backward()                                                     # the forward whose state the plan holds: fine
    # (2): simulate a step that dropped rows: raise the summary word, post it the way train_forward does, next step reports it
    plan = next(iter(cnn.__dict__["_train_plans"].values()))
    assert plan.__dict__["_ovf_state"]["pending"]  # every step posts its word
    plan._check_deferred_overflow()                # ... and a clean one passes
    plan.overflow_any().fill_(1)
    plan._post_deferred_overflow()
    with pytest.raises(RuntimeError, match="exceeded an active-site capacity"):
        cnn(item["voxel_mean"], item["coordinates"], item["batch_size"])
