"""GPU: gradients of the sparse convolution (data + weight, submanifold + strided) vs torch autograd of the
equivalent dense F.conv3d in float64 -- the independent statement used for the forward self-check
(SURVEY.md 8c (i)); then one SECOND train step end to end (finite loss, every parameter receives a gradient)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import assert_features_close, dev
from vision3d_amd import synth
from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


def random_sparse(rng, b, shape, n, c):
    keys = rng.choice(b * shape[0] * shape[1] * shape[2], n, replace=False)
    bb, rem = np.divmod(keys, shape[0] * shape[1] * shape[2])
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    order = np.argsort(bb, kind="stable")
    return np.stack([bb, z, y, x], 1)[order].astype(np.int32), rng.standard_normal((n, c)).astype(np.float32)


def dense_reference(coords, feats, w, b, shape, stride, padding, gout_fn):
    """float64 dense conv3d on the CPU; returns (out rows at given sites fn, dX rows, dW)."""
    x = torch.zeros((b, feats.shape[1], *shape), dtype=torch.float64)
    ft = torch.tensor(feats, dtype=torch.float64, requires_grad=True)
    idx = torch.from_numpy(coords).long()
    xd = x.permute(0, 2, 3, 4, 1).contiguous().index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), ft).permute(0, 4, 1, 2, 3)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = F.conv3d(xd, wt.permute(4, 3, 0, 1, 2), stride=stride, padding=padding)
    return y, ft, wt


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 32), (64, 64), (4, 16)])
def test_subm_gradients(cin, cout):
    from vision3d_amd import spconv
    rng = np.random.default_rng(cin + cout)
    shape, b = [8, 20, 18], 2
    coords, feats = random_sparse(rng, b, shape, 900, cin)
    conv = spconv.SubMConv3d(cin, cout, 3, indice_key="g", bias=False).cuda()
    x = spconv.SparseConvTensor(dev(feats).requires_grad_(True), dev(coords), shape, b)
    out = conv(x)
    gy = rng.standard_normal(out.features.shape).astype(np.float32)
    out.features.backward(dev(gy))
    y, ft, wt = dense_reference(coords, feats, conv.weight.detach().cpu().numpy(), b, shape, 1, 1, None)
    yr = y[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]]
    assert_features_close(out.features.detach().cpu().numpy(), yr.detach().numpy(), "subm forward")
    yr.backward(torch.tensor(gy, dtype=torch.float64))
    assert_features_close(x.features.grad.cpu().numpy(), ft.grad.numpy(), f"subm dX {cin}->{cout}")
    assert_features_close(conv.weight.grad.cpu().numpy(), wt.grad.numpy(), f"subm dW {cin}->{cout}")


@pytest.mark.parametrize("ks,st,pd,cin,cout", [([3, 3, 3], [2, 2, 2], [1, 1, 1], 16, 32), ([3, 3, 3], [2, 2, 2], [0, 1, 1], 64, 64),
                                               ([3, 1, 1], [2, 1, 1], [0, 0, 0], 64, 64)])
def test_strided_gradients(ks, st, pd, cin, cout):
    from vision3d_amd import spconv
    rng = np.random.default_rng(sum(ks) + cin)
    shape, b = [9, 16, 14], 2
    coords, feats = random_sparse(rng, b, shape, 700, cin)
    conv = spconv.SparseConv3d(cin, cout, ks, st, padding=pd, bias=False).cuda()
    x = spconv.SparseConvTensor(dev(feats).requires_grad_(True), dev(coords), shape, b)
    out = conv(x)
    oc = out.indices.cpu().numpy()
    gy = rng.standard_normal(out.features.shape).astype(np.float32)
    out.features.backward(dev(gy))
    y, ft, wt = dense_reference(coords, feats, conv.weight.detach().cpu().numpy(), b, shape, st, pd, None)
    yr = y[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]]
    assert_features_close(out.features.detach().cpu().numpy(), yr.detach().numpy(), "strided forward")
    yr.backward(torch.tensor(gy, dtype=torch.float64))
    assert_features_close(x.features.grad.cpu().numpy(), ft.grad.numpy(), "strided dX")
    assert_features_close(conv.weight.grad.cpu().numpy(), wt.grad.numpy(), "strided dW")


def test_second_train_step_runs_and_is_deterministic():
    """configs[2] shape at bs=2: forward (batch-stat BatchNorm) -> ProposalLoss -> backward -> Adam step."""
    from vision3d_amd.core import Preprocessor, ProposalTargetAssigner
    from vision3d_amd.detector import ProposalLoss, Second
    cfg = second_car_cfg()

    def run():
        torch.manual_seed(0)
        model = Second(cfg).cuda().train()
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
        assigner = ProposalTargetAssigner(cfg)
        items = []
        for s in (0, 1):
            gt = torch.from_numpy(synth.make_gt_boxes(s))
            it = dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long), box_ignore=torch.zeros(len(gt), dtype=torch.bool))
            items.append(assigner(it))
        item = Preprocessor(cfg, seed=0)(dict(points=[synth.make_cloud(0), synth.make_cloud(1)]))
        for k in ("G_cls", "G_reg", "M_cls", "M_reg"):
            item[k] = torch.stack([it[k] for it in items])
        opt.zero_grad()
        losses = ProposalLoss(cfg)(model(item))
        losses["loss"].backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=35)
        opt.step()
        return float(losses["loss"].detach()), grads

    loss_a, grads_a = run()
    loss_b, grads_b = run()
    assert np.isfinite(loss_a) and loss_a > 0
    assert all(g is not None and torch.isfinite(g).all() for g in grads_a.values())
    assert grads_a["cnn.blocks.0.0.0.weight"].abs().sum() > 0 and grads_a["cnn.blocks.3.3.0.weight"].abs().sum() > 0
    # the sparse kernels are deterministic (no atomics): sparse-layer gradients repeat bit for bit
    assert loss_a == loss_b
    for n in ("cnn.blocks.0.0.0.weight", "cnn.blocks.2.1.0.weight", "cnn.blocks.3.3.0.weight"):
        assert torch.equal(grads_a[n], grads_b[n]), n


@pytest.mark.parametrize("n,c,relu", [(5000, 16, True), (70001, 64, True), (333, 32, False), (2, 64, True)])
def test_sparse_batchnorm_relu_matches_torch(n, c, relu):
    """csrc/sparse_bn.hip (training-mode BatchNorm1d + ReLU on sparse features) vs the torch modules: output, input /
    weight / bias gradients and the running statistics after the step."""
    from vision3d_amd.spconv.functional import sparse_batch_norm_relu
    g = torch.Generator().manual_seed(n + c)
    x0 = (torch.randn(n, c, generator=g) * 2.0 + torch.linspace(-3, 3, c)).cuda()
    gy = torch.randn(n, c, generator=g).cuda()

    def make():
        bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, c))
        return bn

    bn_a, bn_b = make(), make()
    xa = x0.clone().requires_grad_(True)
    ya = sparse_batch_norm_relu(xa, bn_a, relu)
    ya.backward(gy)
    xb = x0.clone().requires_grad_(True)
    yb = bn_b(xb)
    if relu:
        yb = torch.relu(yb)
    yb.backward(gy)
    tol = dict(rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(ya.detach().cpu().numpy(), yb.detach().cpu().numpy(), **tol)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(bn_a.weight.grad.cpu().numpy(), bn_b.weight.grad.cpu().numpy(), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(bn_a.bias.grad.cpu().numpy(), bn_b.bias.grad.cpu().numpy(), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(bn_a.running_mean.cpu().numpy(), bn_b.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn_a.running_var.cpu().numpy(), bn_b.running_var.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1
