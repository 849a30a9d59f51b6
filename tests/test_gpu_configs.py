"""GPU: the two BASELINE.json configurations the round-1 suite only touched at reduced size.

configs[4]  Waymo-range sweep (180 000 points, +-75.2 m, 0.05 m voxels): SECOND forward through the native plan.
            * a cropped sub-cloud (same grid, ~25 k points): BEV map against the CPU restatement, feature bar;
            * the full sweep: every stage's active-site list and row count bit-exact against the oracle's rulebook chain,
              no capacity overflow, the large-N kernel's row counts, BEV occupancy == the last stage's sites.
configs[2]  train step at bs = 8 with the bf16-autocast dense half (what bench.py --mode train runs): finite loss,
            sparse gradients repeat bit for bit, one real layer's dX / dW against float64.
"""
import numpy as np
import pytest
import torch

from gpu_util import assert_features_close, assert_fp32_class, numpy_state_dict, randomize_bn
from vision3d_amd import synth
from vision3d_amd.core.config import second_car_cfg, waymo_range_cfg

pytestmark = pytest.mark.gpu


def _waymo_model(seed=0):
    from vision3d_amd.detector import Second
    torch.manual_seed(seed)
    model = Second(waymo_range_cfg())
    randomize_bn(model, seed)
    return model.cuda().eval()


def test_waymo_range_cropped_forward_matches_oracle():
    from oracle import second_cpu
    cfg = waymo_range_cfg()
    model = _waymo_model(1)
    cloud = synth.make_waymo_cloud(0)
    near = cloud[np.abs(cloud[:, :2]).max(1) < 12.0][:25000]  # a dense 24 m x 24 m crop around the sensor, same grid
    assert len(near) > 15000
    with torch.no_grad():
        bev = model.bev_from_points([torch.from_numpy(near).cuda()])
    from oracle import oracle as O
    vox, coords, occ = second_cpu.voxelize_batch([near], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS)
    ref, _, _, _ = second_cpu.sparse_backbone(numpy_state_dict(model), O.vfe_mean(vox, occ), coords,
                                              second_cpu.grid_shape(cfg.GRID_BOUNDS, cfg.VOXEL_SIZE), 1)
    assert bev.shape == ref.shape == (1, 64 * 3, 376, 376)
    ref64 = second_cpu.sparse_backbone64(numpy_state_dict(model), O.vfe_mean(vox, occ), coords,
                                         second_cpu.grid_shape(cfg.GRID_BOUNDS, cfg.VOXEL_SIZE), 1)
    assert_fp32_class(bev.cpu().numpy(), ref, "Waymo-range crop: BEV vs oracle", ref64)


def test_waymo_range_full_sweep_sites_exact_and_large_kernels(oracle):
    from oracle import second_cpu
    cfg = waymo_range_cfg()
    model = _waymo_model(2)
    cloud = synth.make_waymo_cloud(0)
    assert cloud.shape == (180000, 4)
    with torch.no_grad():
        bev = model.bev_from_points([torch.from_numpy(cloud).cuda()])
    plan = next(iter(model._plans.values()))
    torch.cuda.synchronize()
    assert int(plan.overflow().sum().item()) == 0, "a stage hit its capacity at Waymo range"
    # oracle chain: voxel set, then the four strided rulebooks (C, seconds)
    _, coords, _ = second_cpu.voxelize_batch([cloud], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS)
    shape = second_cpu.grid_shape(cfg.GRID_BOUNDS, cfg.VOXEL_SIZE)
    feat, c0, n0, shape0 = plan.layer_output(-1)
    assert shape0 == shape and int(n0.item()) == len(coords) > 100000
    np.testing.assert_array_equal(c0[:len(coords)].cpu().numpy(), coords)
    strided = {2: ([3, 3, 3], [2, 2, 2], [1, 1, 1]), 5: ([3, 3, 3], [2, 2, 2], [1, 1, 1]), 9: ([3, 3, 3], [2, 2, 2], [0, 1, 1]),
               13: ([3, 1, 1], [2, 1, 1], [0, 0, 0])}
    rows = []
    for layer in range(14):
        if layer in strided:
            ks, st, pd = strided[layer]
            coords, _, shape = oracle.sparse_rulebook(coords, shape, ks, st, pd)
        _, c, n, s = plan.layer_output(layer)
        n = int(n.item())
        rows.append(n)
        assert s == shape and n == len(coords), f"layer {layer}: {n} rows vs {len(coords)}"
        if layer in strided:
            np.testing.assert_array_equal(c[:n].cpu().numpy(), coords)  # first-touch order of the output sites, bit-exact
    # the sizes that select the 64-row LDS-shared-weights kernel (>= 32 768 live rows) really occur on this sweep
    assert min(rows[3:9]) >= 32768 and rows[6] >= 32768, rows
    # BEV occupancy == the last stage's sites: a (b, z, y, x) site fills channels [c*D + z] of pixel (y, x)
    occ = (bev[0].reshape(64, shape[0], shape[1], shape[2]) != 0).any(0).cpu().numpy()
    want = np.zeros(shape, bool)
    want[coords[:, 1], coords[:, 2], coords[:, 3]] = True
    assert not (occ & ~want).any()                       # nothing outside the active sites
    assert (occ & want).sum() > 0.97 * want.sum()        # (ReLU can zero all 64 channels of a site, rarely)
    assert torch.isfinite(bev).all()


def test_train_step_bs8_bf16_autocast_is_finite_and_sparse_grads_repeat():
    from vision3d_amd.core import Preprocessor, ProposalTargetAssigner
    from vision3d_amd.detector import ProposalLoss, Second
    cfg = second_car_cfg()
    bs = 8
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in range(bs)]
    assigner = ProposalTargetAssigner(cfg)
    targets = []
    for s in range(bs):
        gt = torch.from_numpy(synth.make_gt_boxes(s))
        targets.append(assigner(dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long),
                                     box_ignore=torch.zeros(len(gt), dtype=torch.bool))))
    tgt = {k: torch.stack([t[k] for t in targets]).cuda() for k in ("G_cls", "G_reg", "M_cls", "M_reg")}

    def run():
        torch.manual_seed(0)
        model = Second(cfg).cuda().train()
        model.rpn = model.rpn.to(memory_format=torch.channels_last)
        model.head = model.head.to(memory_format=torch.channels_last)
        item = Preprocessor(cfg, seed=0)(dict(points=[c.clone() for c in clouds]))
        item.update(tgt)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses = ProposalLoss(cfg)(model(item))
        losses["loss"].backward()
        return float(losses["loss"].detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters()}, item

    loss_a, grads_a, item = run()
    loss_b, grads_b, _ = run()
    assert item["batch_size"] == bs and item["P_cls"].shape == (bs, 1, 2, 200, 176)
    assert np.isfinite(loss_a) and loss_a > 0
    assert all(g is not None and torch.isfinite(g).all() for g in grads_a.values())
    sparse = [n for n in grads_a if n.startswith("cnn.blocks.") and n.endswith(".0.weight")]
    assert len(sparse) == 14
    for n in sparse:  # the sparse backward has no atomics: with identical dense gradients coming in, it repeats bit for bit
        assert grads_a[n].abs().sum() > 0, n
    # the dense half runs the hand-written kernels (fixed-order reductions, no atomics): its gradients repeat bit for bit as well
    dense = [n for n in grads_a if n.startswith(("rpn.", "head."))]
    assert len(dense) >= 25
    for n in dense:
        assert torch.equal(grads_a[n], grads_b[n]), n
    assert loss_a == loss_b
    # the sparse half once more on a FIXED BEV gradient, on its own
    from vision3d_amd import spconv
    torch.manual_seed(0)
    model = Second(cfg).cuda().train()
    g_bev = torch.randn(bs, 128, 200, 176, device="cuda")

    def sparse_half():
        model.zero_grad()
        it = Preprocessor(cfg, seed=0)(dict(points=[c.clone() for c in clouds]))
        bev = model.cnn(it["voxel_mean"], it["coordinates"], it["batch_size"])
        bev.backward(g_bev)
        return {n: p.grad.detach().clone() for n, p in model.cnn.named_parameters()}
    ga, gb = sparse_half(), sparse_half()
    for n in ga:
        assert torch.equal(ga[n], gb[n]), f"sparse gradient of {n} does not repeat"


def test_real_layer_gradients_at_bs8_against_float64():
    """dX / dW of one real backbone layer (stage 2, 64 -> 64 submanifold, ~65 k rows at bs = 8) against float64 sums over the
    same rulebook."""
    from vision3d_amd.core import Preprocessor
    from vision3d_amd import spconv
    from vision3d_amd.spconv.conv import build_sparse_rulebook, build_subm_rulebook
    cfg = second_car_cfg()
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in range(8)]
    item = Preprocessor(cfg, seed=0)(dict(points=clouds))
    x = spconv.SparseConvTensor(item["voxel_mean"], item["coordinates"], [41, 1600, 1408], 8)
    r1 = build_sparse_rulebook(x, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    x1 = spconv.SparseConvTensor(torch.zeros(r1.n, 1, device="cuda"), r1.out_indices, r1.out_shape, 8)
    r2 = build_sparse_rulebook(x1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    n = r2.n
    assert n > 50000
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(n, 64, generator=g).cuda().requires_grad_(True)
    x2 = spconv.SparseConvTensor(feats, r2.out_indices, r2.out_shape, 8)
    conv = spconv.SubMConv3d(64, 64, 3, indice_key="k", bias=False).cuda()
    out = conv(x2)
    gy = torch.randn(n, 64, generator=g).cuda()
    out.features.backward(gy)
    nbr = build_subm_rulebook(x2, [3, 3, 3]).nbr[:, :n].long()  # (27, n)
    w64 = conv.weight.detach().double().reshape(27, 64, 64)
    f64, g64 = feats.detach().double(), gy.double()
    y_ref = torch.zeros(n, 64, dtype=torch.float64, device="cuda")
    dx_ref = torch.zeros_like(y_ref)
    dw_ref = torch.zeros_like(w64)
    for k in range(27):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        i = nbr[k][o]
        y_ref.index_add_(0, o, f64[i] @ w64[k])
        dx_ref.index_add_(0, i, g64[o] @ w64[k].t())
        dw_ref[k] = f64[i].t() @ g64[o]
    assert_features_close(out.features.detach().cpu().numpy(), y_ref.cpu().numpy(), "bs8 forward")
    assert_features_close(feats.grad.cpu().numpy(), dx_ref.cpu().numpy(), "bs8 dX")
    assert_features_close(conv.weight.grad.reshape(27, 64, 64).cpu().numpy(), dw_ref.cpu().numpy(), "bs8 dW")
