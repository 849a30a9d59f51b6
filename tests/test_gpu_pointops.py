"""GPU parity: PV-RCNN point ops (FPS, gather, ball query, group) vs oracle/ -- index outputs exact --
and points-in-boxes vs the reference's golden masks; then the stage-2 feature pieces run end to end."""
import numpy as np
import pytest
import torch

from gpu_util import assert_features_close, dev
from vision3d_amd import synth

pytestmark = pytest.mark.gpu


def test_fps_and_gather_exact(oracle):
    from vision3d_amd.pointnet2.pointnet2_utils import furthest_point_sample, gather_operation
    xyz = np.stack([synth.make_cloud(0)[:, :3], synth.make_cloud(1)[:, :3]])
    idx = furthest_point_sample(dev(xyz), 2048)
    assert idx.dtype == torch.int32 and idx.shape == (2, 2048)
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.fps(xyz, 2048))
    feat = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    np.testing.assert_array_equal(gather_operation(dev(feat), idx).cpu().numpy(), oracle.gather(feat, idx.cpu().numpy()))
    # odd sizes / duplicates (ties -> lowest index) / tiny N
    rng = np.random.default_rng(0)
    for n, k in ((1000, 64), (777, 777), (5, 3), (3000, 100)):
        p = rng.integers(-3, 4, (1, n, 3)).astype(np.float32)  # integer lattice: many exact ties
        np.testing.assert_array_equal(furthest_point_sample(dev(p), k).cpu().numpy(), oracle.fps(p, k))


def test_ball_query_and_group_exact(oracle):
    from vision3d_amd.pointnet2.pointnet2_utils import ball_query, grouping_operation
    rng = np.random.default_rng(1)
    xyz = np.stack([synth.make_cloud(2)[:6000, :3], synth.make_cloud(3)[:6000, :3]])
    new_xyz = np.ascontiguousarray(xyz[:, ::7][:, :700])
    new_xyz[:, :10] += 500.0  # queries with no neighbour at all -> zeros
    for radius, ns in ((0.4, 16), (0.8, 32), (2.4, 16)):
        idx = ball_query(radius, ns, dev(xyz), dev(new_xyz))
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.ball_query(radius, ns, xyz, new_xyz))
    feat = rng.standard_normal((2, 9, 6000)).astype(np.float32)
    np.testing.assert_array_equal(grouping_operation(dev(feat), idx).cpu().numpy(), oracle.group(feat, idx.cpu().numpy()))


def test_points_in_boxes_matches_reference_golden(golden_geom, oracle):
    from vision3d_amd.core.geometry import PointsInCuboids, PointsNotInRectangles, points_in_boxes_mask
    cloud = synth.make_cloud(0)
    boxes = golden_geom["boxes"]
    shape = tuple(golden_geom["mask_shape"])
    n = shape[0] * shape[1]
    for use_z, key in ((True, "mask3d_packed"), (False, "mask2d_packed")):
        ref = np.unpackbits(golden_geom[key])[:n].reshape(shape).astype(bool)
        got = points_in_boxes_mask(dev(cloud), dev(boxes), use_z).cpu().numpy()
        diff = np.argwhere(got != ref)
        print(f"[points_in_boxes] use_z={use_z}: {len(diff)} mismatches of {n}")
        np.testing.assert_array_equal(got, ref)
    per_box = PointsInCuboids(cloud)(boxes)          # numpy in -> list of numpy out (reference contract)
    ref3 = np.unpackbits(golden_geom["mask3d_packed"])[:n].reshape(shape).astype(bool)
    assert len(per_box) == len(boxes)
    for b, pts in enumerate(per_box):
        np.testing.assert_array_equal(pts, cloud[ref3[:, b]])
    ref2 = np.unpackbits(golden_geom["mask2d_packed"])[:n].reshape(shape).astype(bool)
    np.testing.assert_array_equal(PointsNotInRectangles(cloud)(boxes), cloud[~ref2.any(1)])


def test_sa_module_matches_cpu_composition(oracle):
    """PointnetSAModuleMSG (VSA building block): device ops + torch MLP vs oracle ops + the same MLP on CPU."""
    from vision3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(3)
    sa = PointnetSAModuleMSG(npoint=-1, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[6, 16, 16], [6, 16, 32]], use_xyz=True).eval()
    rng = np.random.default_rng(2)
    xyz = synth.make_cloud(6)[None, :5000, :3]
    feat = rng.standard_normal((1, 6, 5000)).astype(np.float32)
    new_xyz = np.ascontiguousarray(xyz[:, ::10])
    outs = []
    for (radius, ns), mlp in zip(((0.8, 16), (1.6, 32)), sa.mlps):
        idx = oracle.ball_query(radius, ns, xyz, new_xyz)
        g_xyz = oracle.group(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
        g = np.concatenate([g_xyz, oracle.group(feat, idx)], 1)
        with torch.no_grad():
            outs.append(mlp(torch.from_numpy(g)).max(3).values.numpy())
    ref = np.concatenate(outs, 1)
    sa = sa.cuda()
    with torch.no_grad():
        _, got = sa(dev(xyz), dev(feat), dev(new_xyz))
    assert_features_close(got.cpu().numpy(), ref, "SA-MSG")


@pytest.mark.parametrize("kf,nout,ns", [(4, 16, 16), (4, 16, 32), (32, 32, 16), (64, 64, 32), (512, 192, 16), (512, 192, 32), (192, 96, 32)])
def test_sa_mlp_layer_matches_fp64(kf, nout, ns):
    """csrc/sa_mlp.hip: gathered first layer (xyz offsets + point-major features), plain later layer, with and without the
    max over the samples, against the same sums in float64 (exact-fp32 MFMA: only accumulation order differs)."""
    from vision3d_amd.pointnet2.pointnet2_utils import ball_query, sa_mlp_layer
    rng = np.random.default_rng(kf + nout + ns)
    b, n, m = 2, 3000, 333
    xyz = np.stack([synth.make_cloud(4)[:n, :3], synth.make_cloud(5)[:n, :3]])
    new_xyz = np.ascontiguousarray(xyz[:, ::9][:, :m])
    feat = rng.standard_normal((b, n, kf)).astype(np.float32)
    w = (rng.standard_normal((4 + kf, nout)) / np.sqrt(4 + kf)).astype(np.float32)
    w[3] = 0
    bias = rng.standard_normal(nout).astype(np.float32) * 0.1
    idx = ball_query(1.2, ns, dev(xyz), dev(new_xyz))
    ii = idx.cpu().numpy().astype(np.int64)
    bidx = np.arange(b)[:, None, None]
    rel = xyz[bidx, ii].astype(np.float64) - new_xyz[:, :, None, :].astype(np.float32).astype(np.float64)
    rel32 = (xyz[bidx, ii] - new_xyz[:, :, None, :]).astype(np.float64)  # the kernel subtracts in float32
    rows = np.concatenate([rel32, np.zeros(rel.shape[:3] + (1,)), feat[bidx, ii].astype(np.float64)], -1).reshape(-1, 4 + kf)
    for relu in (True, False):
        ref = rows @ w.astype(np.float64) + bias
        if relu:
            ref = np.maximum(ref, 0)
        got = sa_mlp_layer(dev(feat), dev(w), dev(bias), relu, False, xyz=dev(xyz), new_xyz=dev(new_xyz), idx=idx).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
        gotp = sa_mlp_layer(dev(feat), dev(w), dev(bias), relu, True, xyz=dev(xyz), new_xyz=dev(new_xyz), idx=idx).cpu().numpy()
        np.testing.assert_allclose(gotp, ref.reshape(b * m, ns, nout).max(1), rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    # later layer: identity rows
    x = rng.standard_normal((b * m * ns, kf)).astype(np.float32)
    w2 = (rng.standard_normal((kf, nout)) / np.sqrt(kf)).astype(np.float32)
    ref = np.maximum(x.astype(np.float64) @ w2.astype(np.float64) + bias, 0)
    got = sa_mlp_layer(dev(x), dev(w2), dev(bias), True, False, groups=(b, m, ns)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    gotp = sa_mlp_layer(dev(x), dev(w2), dev(bias), True, True, groups=(b, m, ns)).cpu().numpy()
    np.testing.assert_allclose(gotp, ref.reshape(b * m, ns, nout).max(1), rtol=2e-5, atol=2e-5 * np.abs(ref).max())


def test_sa_module_fused_equals_torch_path_and_is_differentiable():
    """Inference runs the fused kernels; with autograd on the module takes the torch path (grouping_operation is
    differentiable: scatter-add backward) -- same features, and the gradient reaches the point features."""
    from gpu_util import randomize_bn
    from vision3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(4)
    sa = PointnetSAModuleMSG(npoint=-1, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[5, 8, 16], [5, 32, 64]], use_xyz=True)
    randomize_bn(sa, 1)
    sa = sa.cuda().eval()
    xyz = dev(synth.make_cloud(7)[None, :4000, :3])
    feat = torch.randn(1, 5, 4000, device="cuda", requires_grad=True)
    new_xyz = xyz[:, ::8].contiguous()
    with torch.no_grad():
        _, fused = sa(xyz, feat.detach(), new_xyz)
        _, fused_pm = sa(xyz, None, new_xyz, features_pm=feat.detach().transpose(1, 2).contiguous())
    _, ref = sa(xyz, feat, new_xyz)  # autograd on -> torch path
    assert torch.equal(fused, fused_pm)
    assert_features_close(fused.cpu().numpy(), ref.detach().cpu().numpy(), "fused SA vs torch path")
    ref.square().sum().backward()
    assert feat.grad is not None and torch.isfinite(feat.grad).all() and float(feat.grad.abs().sum()) > 0
    # the gradient of the gather is a scatter-add: check against the dense statement on a small case
    from vision3d_amd.pointnet2.pointnet2_utils import gather_operation, grouping_operation
    f = torch.randn(2, 3, 50, device="cuda", dtype=torch.float32, requires_grad=True)
    gi = torch.randint(0, 50, (2, 7, 4), device="cuda", dtype=torch.int32)
    out = grouping_operation(f, gi)
    wgt = torch.randn_like(out)
    (out * wgt).sum().backward()
    dense = torch.zeros_like(f)
    for bb in range(2):
        for mm in range(7):
            for ss in range(4):
                dense[bb, :, gi[bb, mm, ss]] += wgt[bb, :, mm, ss]
    torch.testing.assert_close(f.grad, dense, rtol=1e-5, atol=1e-5)
    f.grad = None
    k = torch.randint(0, 50, (2, 9), device="cuda", dtype=torch.int32)
    out = gather_operation(f, k)
    wgt = torch.randn_like(out)
    (out * wgt).sum().backward()
    dense = torch.zeros_like(f)
    for bb in range(2):
        for kk in range(9):
            dense[bb, :, k[bb, kk]] += wgt[bb, :, kk]
    torch.testing.assert_close(f.grad, dense, rtol=1e-5, atol=1e-5)


def _sa_cpu(oracle, sa_cpu, xyz, feat_cn, new_xyz):
    """PointnetSAModuleMSG by hand: oracle ball query / group + the module's own MLPs on the CPU."""
    outs = []
    for grouper, mlp in zip(sa_cpu.groupers, sa_cpu.mlps):
        idx = oracle.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
        g_xyz = oracle.group(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
        g = np.concatenate([g_xyz, oracle.group(np.ascontiguousarray(feat_cn), idx)], 1)
        with torch.no_grad():
            outs.append(mlp(torch.from_numpy(g)).max(3).values.numpy())
    return np.concatenate(outs, 1)


def test_vsa_and_roi_grid_pool_match_cpu_composition(oracle):
    """The five voxel-set-abstraction levels, the BEV gather and RoI-grid pooling of PV-RCNN on real stage-1 outputs against
    the composition  oracle ball query / group -> the modules' own weights on the CPU -> max  (injected grid samples)."""
    import copy
    from gpu_util import randomize_bn
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = PV_RCNN(cfg)
    randomize_bn(model, 2)
    cpu = copy.deepcopy(model).eval()
    model = model.cuda().eval()
    item = Preprocessor(cfg, seed=0)(dict(points=[synth.make_cloud(0)[:6000]], anchors=AnchorGenerator(cfg).anchors.cuda()))
    with torch.no_grad():
        item = model.proposal(item)
        kp = item["keypoints"]
        xyz, refl = item["points"].split([3, 1], dim=-1)
        sources = [(xyz, refl), *item["_cnn_features"]]
        got = model._pointnets(sources, kp)
        kp_np = kp.cpu().numpy()
        for lvl, (pnet_cpu, (sx, sf), g) in enumerate(zip(cpu.pnets, sources, got)):
            ref = _sa_cpu(oracle, pnet_cpu, sx.cpu().numpy(), sf.transpose(1, 2).cpu().numpy(), kp_np)
            assert_features_close(g.cpu().numpy(), ref, f"VSA level {lvl}")
        pf = model.point_feature_extract(item, item["_cnn_features"], item["_bev_map"])
        bev_ref = cpu.bev(item["_bev_map"].cpu(), kp.cpu()).numpy()
        assert_features_close(pf[:, -bev_ref.shape[1]:].cpu().numpy(), bev_ref, "BEV gather")
        props = torch.from_numpy(synth.make_gt_boxes(0)[None, :24]).cuda()
        samples = torch.rand((1, 24, cfg.GRIDPOOL.NUM_GRIDPOINTS, 3), generator=torch.Generator().manual_seed(5))
        pooled = model.roi_grid_pool(props, kp, pf, samples.cuda())
        pts = cpu.roi_grid_pool.sample_gridpoints(props.cpu(), samples).reshape(1, -1, 3).numpy()
        sa_ref = _sa_cpu(oracle, cpu.roi_grid_pool.pnet, kp_np, pf.cpu().numpy(), np.ascontiguousarray(pts))
        per_box = torch.from_numpy(sa_ref).reshape(1, -1, 24, cfg.GRIDPOOL.NUM_GRIDPOINTS).permute(0, 2, 1, 3).reshape(1, 24, -1)
        ref = cpu.roi_grid_pool.reduction(per_box).numpy()
        assert_features_close(pooled.cpu().numpy(), ref, "RoI-grid pool")


def test_pv_rcnn_forward_and_inference():
    """PV_RCNN.forward / .inference (stubs upstream, model.py:84-85): shapes, reproducible with injected grid samples, the
    refined boxes are box_encode.decode of the residuals, and the proposals are stage 1's top-k decoded anchors."""
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    from vision3d_amd.core.box_encode import decode
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    cfg = second_car_cfg()
    torch.manual_seed(1)
    model = PV_RCNN(cfg).cuda().eval()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    clouds = [synth.make_cloud(0), synth.make_cloud(1)[:15000]]
    n = cfg.NUM_CLASSES * cfg.PROPOSAL.TOPK
    samples = torch.rand((2, n, cfg.GRIDPOOL.NUM_GRIDPOINTS, 3), generator=torch.Generator().manual_seed(2)).cuda()
    def seeded():  # the per-frame padding of the sparse levels draws random rows (sparse_cnn.py:33-37): seed it
        model.cnn.pad_generator = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        seeded()
        a = model(Preprocessor(cfg, seed=0)(dict(points=clouds, anchors=anchors)), samples)
        seeded()
        b = model(Preprocessor(cfg, seed=0)(dict(points=clouds, anchors=anchors)), samples)
        dets = model.inference(Preprocessor(cfg, seed=0)(dict(points=clouds, anchors=anchors)), samples)
    assert a["proposals"].shape == (2, n, 7) and a["pooled_features"].shape == (2, n, 256)
    assert a["R_reg"].shape == (2, n, 7) and a["R_cls"].shape == (2, n, 1) and a["keypoint_features"].shape == (2, 512, 2048)
    for k in ("proposals", "pooled_features", "R_reg", "R_cls", "boxes_refined"):
        assert torch.equal(a[k], b[k]), k
    torch.testing.assert_close(a["boxes_refined"], decode(a["R_reg"], a["proposals"]))
    head = model.proposal_layer
    sc, ai = a["P_cls"].sigmoid().reshape(2, cfg.NUM_CLASSES, -1).topk(head.TOPK, -1)
    torch.testing.assert_close(a["proposal_scores"], sc.reshape(2, -1))
    assert (a["proposal_scores"][:, :-1] >= a["proposal_scores"][:, 1:]).all()
    boxes, bidx, cidx, scores = dets
    assert boxes.shape[1] == 7 and len(boxes) == len(bidx) == len(cidx) == len(scores)
    assert (scores[:-1] >= scores[1:]).all() and set(bidx.tolist()) <= {0, 1}


def test_pv_rcnn_prefetched_keypoints_change_nothing():
    """PV_RCNN.prefetch_keypoints (the next frame's farthest-point sampling started on the side stream before the current frame is
    issued) against the plain call: same keypoints, same detections, over a few frames in flight."""
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    cfg = second_car_cfg()
    torch.manual_seed(4)
    model = PV_RCNN(cfg).cuda().eval()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    n = cfg.NUM_CLASSES * cfg.PROPOSAL.TOPK
    samples = torch.rand((1, n, cfg.GRIDPOOL.NUM_GRIDPOINTS, 3), generator=torch.Generator().manual_seed(5)).cuda()
    clouds = [synth.make_cloud(10 + i) for i in range(4)]
    make = lambda i: Preprocessor(cfg, seed=0)(dict(points=[clouds[i]], anchors=anchors))

    def seeded():
        model.cnn.pad_generator = torch.Generator(device="cuda").manual_seed(6)
    with torch.no_grad():
        plain = []
        for i in range(4):
            seeded()
            item = make(i)
            plain.append((model.inference(item, samples), item["keypoints"].clone()))
        item = model.prefetch_keypoints(make(0))
        assert "_keypoints_ready" in item
        for i in range(4):
            nxt = model.prefetch_keypoints(make(i + 1)) if i < 3 else None
            seeded()
            dets = model.inference(item, samples)
            assert "_keypoints_ready" not in item
            assert torch.equal(item["keypoints"], plain[i][1])
            for x, y in zip(dets, plain[i][0]):
                assert torch.equal(x, y)
            item = nxt
    torch.cuda.synchronize()


@pytest.mark.parametrize("batch", [1, 2])
def test_pv_rcnn_native_cnn_equals_the_module_path(batch):
    """Stage 1 of an inference frame through the backbone plan (PV_RCNN._native_cnn) against the module-by-module sparse CNN: the
    same sites in the same order at every level (exact), features and BEV map equal up to the f16s rounding of a calibrated instead
    of a per-call activation scale, and detections that agree."""
    from gpu_util import FP32_CLASS_FLOOR, assert_features_close
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    cfg = second_car_cfg()
    torch.manual_seed(7)
    model = PV_RCNN(cfg).cuda().eval()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    clouds = [synth.make_cloud(20 + i)[: 16384 - 900 * i] for i in range(batch)]
    make = lambda: Preprocessor(cfg, seed=0)(dict(points=clouds, anchors=anchors))
    results = {}
    with torch.no_grad():
        for native in (False, True, True):  # twice natively: the second frame runs on the calibrated plan without its set-up passes
            model.native_cnn = native
            model.cnn.pad_generator = torch.Generator(device="cuda").manual_seed(8)
            item = model.proposal(make())
            results[native] = ([(x.clone(), f.clone()) for x, f in item["_cnn_features"]], item["_bev_map"].clone(), item["P_cls"].clone())
    (lv_a, bev_a, cls_a), (lv_b, bev_b, cls_b) = results[False], results[True]
    assert len(lv_a) == len(lv_b) == 4
    for level, ((xa, fa), (xb, fb)) in enumerate(zip(lv_a, lv_b)):
        assert xa.shape == xb.shape and fa.shape == fb.shape, (level, xa.shape, xb.shape)
        assert torch.equal(xa, xb), f"level {level}: the sites differ"
        assert_features_close(fb.cpu().numpy().reshape(-1, fb.shape[-1]), fa.cpu().numpy().reshape(-1, fa.shape[-1]),
                              f"level {level} features", floor=FP32_CLASS_FLOOR)
    assert_features_close(bev_b.cpu().numpy().reshape(batch, -1), bev_a.cpu().numpy().reshape(batch, -1), "BEV map", floor=FP32_CLASS_FLOOR)
    assert float((cls_a - cls_b).abs().max()) < 1e-4 * max(1.0, float(cls_a.abs().max()))


@pytest.mark.parametrize("batch", [1, 2])
def test_pv_rcnn_native_proposal_tail_equals_the_torch_statements(batch):
    """PV_RCNN.native_tail: stage-1 top-k + decode (v3d_proposals_topk) and the stage-2 tail -- refined-box decode, sigmoid, batched
    rotated NMS, score cut (v3d_refine_nms) -- against the torch statements of the same steps on the same frame: same proposals in
    the same order, same survivors in the same order, boxes and scores within float rounding (device expf vs torch's)."""
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    cfg = second_car_cfg()
    torch.manual_seed(21)
    model = PV_RCNN(cfg).cuda().eval()
    with torch.no_grad():  # scores that straddle the class threshold and overlapping boxes: both cuts must have work to do
        model.proposal_layer.conv_cls.bias.fill_(0.3)
        model.refinement_layer.mlp[-1].bias[7] = 0.2
        model.refinement_layer.mlp[-1].weight.mul_(30.0)
    anchors = AnchorGenerator(cfg).anchors.cuda()
    clouds = [synth.make_cloud(30 + i) for i in range(batch)]
    n = cfg.NUM_CLASSES * cfg.PROPOSAL.TOPK
    samples = torch.rand((batch, n, cfg.GRIDPOOL.NUM_GRIDPOINTS, 3), generator=torch.Generator().manual_seed(22)).cuda()
    outs = {}
    with torch.no_grad():
        for native in (True, False):
            model.native_tail = native
            model.cnn.pad_generator = torch.Generator(device="cuda").manual_seed(23)
            item = Preprocessor(cfg, seed=0)(dict(points=clouds, anchors=anchors))
            dets = model.inference(item, samples)
            outs[native] = (dets, item["proposals"].clone(), item["proposal_scores"].clone(), item["boxes_refined"].clone())
    (da, pa, sa, ra), (db, pb, sb, rb) = outs[True], outs[False]
    assert pa.shape == pb.shape == (batch, n, 7)
    torch.testing.assert_close(sa, sb, rtol=0, atol=0)
    torch.testing.assert_close(pa, pb, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ra, rb, rtol=1e-5, atol=1e-5)
    assert len(da[0]) == len(db[0]) and 0 < len(da[0]) < batch * n, (len(da[0]), len(db[0]))
    assert torch.equal(da[1], db[1]) and torch.equal(da[2], db[2])
    torch.testing.assert_close(da[3], db[3], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(da[0], db[0], rtol=1e-5, atol=1e-5)
    assert (da[3][:-1] >= da[3][1:]).all()


def test_pv_rcnn_stage_pieces_run():
    """configs[3] shapes: FPS keypoints + 5-level VSA + BEV gather -> (B, 512, 2048); RoI-grid pool -> (B, n, 256)."""
    from vision3d_amd.core import Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = PV_RCNN(cfg).cuda().eval()
    item = Preprocessor(cfg, seed=0)(dict(points=[synth.make_cloud(0), synth.make_cloud(1)[:15000]]))
    with torch.no_grad():
        item = model.proposal(item)
        assert item["keypoints"].shape == (2, 2048, 3)
        assert item["P_cls"].shape == (2, 1, 2, 200, 176)
        pf = model.point_feature_extract(item, item["_cnn_features"], item["_bev_map"])
        assert pf.shape == (2, 512, 2048) and torch.isfinite(pf).all()
        props = torch.from_numpy(np.stack([synth.make_gt_boxes(0)[:20], synth.make_gt_boxes(1)[:20]])).cuda()
        pooled = model.roi_grid_pool(props, item["keypoints"], pf)
        assert pooled.shape == (2, 20, 256)
        deltas, conf = model.refinement_layer(None, pooled, props)
        assert deltas.shape == (2, 20, 7) and conf.shape == (2, 20, 1)


@pytest.mark.parametrize("n,m,ra,nsa,rb,nsb", [(16384, 2048, 0.4, 16, 0.8, 16), (4204, 2048, 2.4, 16, 4.8, 32), (777, 130, 1.0, 5, 3.0, 64),
                                               (2500, 37, 0.01, 16, 100.0, 32)])
def test_ball_query_pair_equals_two_single_queries(n, m, ra, nsa, rb, nsb):
    """v3d_ball_query (two radii in one scan) == v3d_ball_query per radius (itself exact against the oracle above): ragged
    sizes, a radius that finds nothing, one that finds everything."""
    from vision3d_amd.pointnet2 import pointnet2_utils as PU
    g = torch.Generator().manual_seed(n + m)
    xyz = (torch.rand(2, n, 3, generator=g) * torch.tensor([70.0, 80.0, 4.0])).cuda()
    new_xyz = xyz[:, torch.randperm(n, generator=g)[:m]].contiguous() + 0.05
    ia, ib = PU.ball_query_pair(ra, nsa, rb, nsb, xyz, new_xyz)
    assert torch.equal(ia, PU.ball_query(ra, nsa, xyz, new_xyz))
    assert torch.equal(ib, PU.ball_query(rb, nsb, xyz, new_xyz))


def _both_algos(fn):
    from vision3d_amd.pointnet2 import pointnet2_utils as PU
    out = {}
    for algo in ("scan", "grid"):
        old, PU.BALL_QUERY_ALGO = PU.BALL_QUERY_ALGO, algo
        try:
            out[algo] = fn(PU)
        finally:
            PU.BALL_QUERY_ALGO = old
    return out["scan"], out["grid"]


@pytest.mark.parametrize("order", ["shuffled", "scan"])
def test_ball_query_grid_equals_scan_at_pvrcnn_sizes(order):
    """v3d_ball_query_grid == v3d_ball_query bit for bit on the six calls of a PV-RCNN frame's shapes: raw cloud and voxel-centre
    databases around 2 048 keypoints (radii of config.py PSA.RADII), the keypoints around RoI grid points."""
    from vision3d_amd.pointnet2 import pointnet2_utils as PU
    cloud = torch.from_numpy(synth.make_cloud(5, order=order)[:, :3]).cuda()[None].contiguous()
    kp = cloud[:, PU.furthest_point_sample(cloud, 2048)[0].long()].contiguous()
    vs = torch.tensor([0.05, 0.05, 0.1], device="cuda")
    databases = [(cloud, (0.4, 0.8))]
    for stride, radii in zip((1, 2, 4, 8), ((0.4, 0.8), (0.8, 1.2), (1.2, 2.4), (2.4, 4.8))):
        cells = torch.unique(torch.floor(cloud[0] / (vs * stride)), dim=0)
        cells = cells[torch.randperm(cells.shape[0], generator=torch.Generator().manual_seed(stride)).cuda()]
        databases.append(((cells * (vs * stride))[None].contiguous(), radii))
    for db, (ra, rb) in databases:
        (sa, sb), (ga, gb) = _both_algos(lambda PU: PU.ball_query_pair(ra, 16, rb, 32, db, kp))
        assert torch.equal(sa, ga) and torch.equal(sb, gb), (db.shape, ra, rb)
        assert (ga != 0).any() and (gb != 0).any()
    grid_pts = (kp[:, :1600] + 0.3 * torch.randn(1, 1600, 3, generator=torch.Generator().manual_seed(3)).cuda()).contiguous()
    (sa, sb), (ga, gb) = _both_algos(lambda PU: PU.ball_query_pair(0.8, 16, 1.6, 32, kp, grid_pts))
    assert torch.equal(sa, ga) and torch.equal(sb, gb)


@pytest.mark.parametrize("n,m,ra,nsa,rb,nsb", [(16384, 2048, 0.4, 16, 0.8, 16), (4204, 2048, 2.4, 16, 4.8, 32), (777, 130, 1.0, 5, 3.0, 64),
                                               (2500, 37, 0.01, 16, 100.0, 32), (1, 9, 1.0, 4, 2.0, 4), (65, 3, 0.5, 100, 0.6, 70),
                                               (40000, 500, 0.3, 16, 0.2, 32)])
def test_ball_query_grid_equals_scan_ragged_and_degenerate(n, m, ra, nsa, rb, nsb):
    """Ragged sizes, a radius that finds nothing / everything, nsample beyond a wave, the larger radius given first, two frames; then
    the corner cases of the binning: non-finite database points and queries, queries far outside the database, a database of
    coincident points, a database spread so wide that the cells must grow past the radius."""
    g = torch.Generator().manual_seed(n * 7 + m)
    xyz = (torch.rand(2, n, 3, generator=g) * torch.tensor([70.0, 80.0, 4.0])).cuda()
    new_xyz = xyz[:, torch.randint(0, n, (m,), generator=g)].contiguous() + 0.05
    (sa, sb), (ga, gb) = _both_algos(lambda PU: PU.ball_query_pair(ra, nsa, rb, nsb, xyz, new_xyz))
    assert torch.equal(sa, ga) and torch.equal(sb, gb)
    s1, g1 = _both_algos(lambda PU: PU.ball_query(rb, nsb, xyz, new_xyz))
    assert torch.equal(s1, g1) and torch.equal(g1, gb)
    # non-finite coordinates, far-away queries
    bad = xyz.clone()
    k = max(n // 10, 1)
    bad[0, :k, 0] = float("nan")
    bad[1, :k, 1] = float("inf")
    bad[0, -k:, 2] = float("nan")
    q = new_xyz.clone()
    q[0, 0] = float("nan")
    q[1, 0, 1] = float("-inf")
    q[0, 1 % m] = 1e6
    q[1, 2 % m, 0] = -3.0  # just outside the database's bounds
    (sa, sb), (ga, gb) = _both_algos(lambda PU: PU.ball_query_pair(ra, nsa, rb, nsb, bad, q))
    assert torch.equal(sa, ga) and torch.equal(sb, gb)
    # coincident points; a very wide database (cells grown beyond the radius)
    same = torch.full_like(xyz, 2.5)
    (sa, sb), (ga, gb) = _both_algos(lambda PU: PU.ball_query_pair(ra, nsa, rb, nsb, same, new_xyz))
    assert torch.equal(sa, ga) and torch.equal(sb, gb)
    wide = xyz * torch.tensor([300.0, 300.0, 1.0], device="cuda")
    qw = wide[:, torch.randint(0, n, (m,), generator=g)].contiguous() + 0.01
    (sa, sb), (ga, gb) = _both_algos(lambda PU: PU.ball_query_pair(ra, nsa, rb, nsb, wide, qw))
    assert torch.equal(sa, ga) and torch.equal(sb, gb)


def test_ball_query_grids_of_several_databases_in_one_launch():
    """PU.ball_query_grids: the grids of several databases (different sizes, different radii, two frames) from ONE launch, each then
    queried more than once (any radius up to the one it was built for) -- equal to the scan kernel; a grid refuses another tensor, a
    changed tensor and a larger radius."""
    from vision3d_amd.pointnet2 import pointnet2_utils as PU
    g = torch.Generator().manual_seed(11)
    dbs = [(torch.rand(2, n, 3, generator=g) * torch.tensor([70.0, 80.0, 4.0])).cuda() for n in (16384, 13001, 4612, 2048, 1, 300, 9000)]
    radii = [0.8, 0.8, 4.8, 1.6, 1.0, 2.0, 2.4]
    grids = PU.ball_query_grids(list(zip(dbs, radii)))
    q = dbs[0][:, torch.randint(0, 16384, (700,), generator=g)].contiguous() + 0.03
    for db, r, grid in zip(dbs, radii, grids):
        for ra, rb in ((r / 2, r), (r, r / 3)):
            old, PU.BALL_QUERY_ALGO = PU.BALL_QUERY_ALGO, "scan"
            try:
                sa, sb = PU.ball_query_pair(ra, 16, rb, 32, db, q)
            finally:
                PU.BALL_QUERY_ALGO = old
            ga, gb = PU.ball_query_pair(ra, 16, rb, 32, db, q, grid=grid)
            assert torch.equal(sa, ga) and torch.equal(sb, gb), (db.shape, ra, rb)
            assert torch.equal(PU.ball_query(rb, 32, db, q, grid=grid), gb)
    with pytest.raises(RuntimeError):
        PU.ball_query(radii[0] * 1.5, 16, dbs[0], q, grid=grids[0])
    with pytest.raises(RuntimeError):
        PU.ball_query(radii[0], 16, dbs[0].clone(), q, grid=grids[0])
    dbs[0].add_(1.0)
    with pytest.raises(RuntimeError):
        PU.ball_query(radii[0], 16, dbs[0], q, grid=grids[0])


def test_many_queries_and_many_products_in_one_launch_equal_the_single_calls():
    """PU.ball_query_pairs_many (several databases, the same queries, one launch) == ball_query_pair per database; PU.linear_rows_many
    == linear_rows per product (different shapes in one launch)."""
    from vision3d_amd.pointnet2 import pointnet2_utils as PU
    g = torch.Generator().manual_seed(21)
    dbs = [(torch.rand(2, n, 3, generator=g) * torch.tensor([70.0, 80.0, 4.0])).cuda() for n in (16384, 13001, 17000, 4612, 300)]
    radii = [(0.4, 0.8), (0.4, 0.8), (0.8, 1.2), (2.4, 4.8), (1.0, 0.5)]
    grids = PU.ball_query_grids([(d, max(r)) for d, r in zip(dbs, radii)])
    q = dbs[0][:, torch.randint(0, 16384, (2048,), generator=g)].contiguous() + 0.03
    outs = PU.ball_query_pairs_many([(gr, d, ra, 16, rb, 32) for gr, d, (ra, rb) in zip(grids, dbs, radii)], q)
    for (ia, ib), gr, d, (ra, rb) in zip(outs, grids, dbs, radii):
        sa, sb = PU.ball_query_pair(ra, 16, rb, 32, d, q, grid=gr)
        assert torch.equal(ia, sa) and torch.equal(ib, sb)
    jobs = [(torch.randn(r, k, device="cuda"), torch.randn(k, n, device="cuda") / k ** 0.5)
            for r, k, n in ((16384, 4, 32), (13001, 16, 16), (777, 64, 128), (100, 3072, 256), (1, 4, 16))]
    for got, (a, w) in zip(PU.linear_rows_many(jobs), jobs):
        assert torch.equal(got, PU.linear_rows(a, w))


def test_ball_query_grid_workspace_is_checked():
    from vision3d_amd import _lib as L
    xyz = torch.rand(1, 100, 3).cuda()
    idx = torch.empty((1, 100, 4), dtype=torch.int32, device="cuda")
    need = L.lib().v3d_ball_query_grid_workspace(1, 100)
    assert need >= 100 * 16
    small = torch.empty(need - 16, dtype=torch.uint8, device="cuda")
    rc = L.lib().v3d_ball_query_grid(L.ptr(xyz), L.ptr(xyz), 1, 100, 100, 1.0, 4, L.ptr(idx), 0.0, 0, None, L.ptr(small), small.numel(),
                                     L.stream_ptr())
    assert rc != 0


@pytest.mark.parametrize("r,k,nout,bias,relu", [(100, 3072, 256, False, True), (100, 256, 256, False, True), (100, 256, 128, True, True),
                                                (100, 128, 8, True, False), (1, 4, 16, True, False), (37, 260, 40, False, False),
                                                (300, 1028, 96, True, True), (10540, 64, 128, False, False), (16384, 4, 32, True, True),
                                                (2048, 512, 384, False, False), (777, 36, 48, True, False)])
def test_linear_rows_matches_fp64(r, k, nout, bias, relu):
    """csrc/sa_mlp.hip linear_rows_kernel (the reduction / refinement MLP layers of PV-RCNN: a hundred rows, K split over the waves)
    against the same sums in float64; strided input rows, a column block as output, an output width that is not a multiple of 16."""
    from vision3d_amd.pointnet2.pointnet2_utils import linear_rows
    rng = np.random.default_rng(r + k + nout)
    npad = -(-nout // 16) * 16
    a = rng.standard_normal((r, k + 8)).astype(np.float32)
    w = np.zeros((k, npad), np.float32)
    w[:, :nout] = rng.standard_normal((k, nout)) / np.sqrt(k)
    bv = np.zeros(npad, np.float32)
    bv[:nout] = rng.standard_normal(nout) * 0.1
    ref = a[:, :k].astype(np.float64) @ w[:, :nout].astype(np.float64) + (bv[:nout] if bias else 0)
    if relu:
        ref = np.maximum(ref, 0)
    ad = dev(a)[:, :k]  # row stride k + 8
    got = linear_rows(ad, dev(w), dev(bv) if bias else None, relu, n_store=nout)
    assert got.shape == (r, nout)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    wide = torch.full((r, nout + 7), -7.0, device="cuda")
    linear_rows(ad, dev(w), dev(bv) if bias else None, relu, out=wide[:, 3:3 + nout], n_store=nout)
    assert torch.equal(wide[:, 3:3 + nout], got) and (wide[:, :3] == -7).all() and (wide[:, 3 + nout:] == -7).all()
    assert torch.equal(linear_rows(ad, dev(w), dev(bv) if bias else None, relu, n_store=nout), got)  # fixed summation order


def test_mlp_native_forward_matches_the_torch_modules():
    """detector/layers.py MLP: eval + no_grad runs v3d_linear_rows per layer (bias / ReLU in the epilogue), otherwise nn.Sequential;
    the permuted first layer of RoiGridPool equals permuting the activations."""
    from vision3d_amd.detector.layers import MLP
    torch.manual_seed(3)
    for channels, kw in (([3072, 256, 256], {}), ([256, 128, 8], dict(bias=True, relu=[True, False]))):
        mlp = MLP(channels, **kw).cuda().eval()
        for lin in (m for m in mlp if isinstance(m, torch.nn.Linear)):
            torch.nn.init.normal_(lin.weight, std=channels[0] ** -0.5)
            if lin.bias is not None:
                torch.nn.init.normal_(lin.bias, std=0.1)
        x = torch.randn(2, 50, channels[0], device="cuda")
        with torch.no_grad():
            got = mlp(x)
            assert mlp.native_ok(x)
            ref = torch.nn.Sequential.forward(mlp.double(), x.double())
            mlp.float()
        assert got.shape == ref.shape
        torch.testing.assert_close(got.double(), ref, rtol=2e-5, atol=2e-5 * ref.abs().max().item())
        x.requires_grad_(True)
        assert not mlp.native_ok(x.detach()) or torch.is_grad_enabled()
        mlp(x).sum().backward()  # autograd on: the torch modules
        assert x.grad is not None
    mlp = MLP([64, 32, 16]).cuda().eval()
    x = torch.randn(7, 64, device="cuda")
    perm = torch.randperm(64, device="cuda")
    with torch.no_grad():
        torch.testing.assert_close(mlp.native_forward(x[:, perm], first_rows=perm), mlp.native_forward(x), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kf,n1,n2,ns", [(4, 16, 16, 16), (32, 32, 32, 32), (64, 64, 64, 16), (512, 192, 96, 32), (512, 192, 96, 16)])
def test_sa_mlp_pair_matches_fp64_and_the_two_layer_path(kf, n1, n2, ns):
    """csrc/sa_mlp.hip PAIR: feature part of the first layer once per database point (linear_rows) + v3d_sa_mlp_pair, against float64
    and against the launch-per-layer path (equal up to the summation order of the first layer)."""
    from vision3d_amd.pointnet2.pointnet2_utils import ball_query, linear_rows, sa_mlp_layer, sa_mlp_pair
    rng = np.random.default_rng(kf + n1 + ns)
    b, n, m = 2, 3000, 333
    xyz = np.stack([synth.make_cloud(4)[:n, :3], synth.make_cloud(5)[:n, :3]])
    new_xyz = np.ascontiguousarray(xyz[:, ::9][:, :m])
    feat = rng.standard_normal((b, n, kf)).astype(np.float32)
    w1 = (rng.standard_normal((4 + kf, n1)) / np.sqrt(4 + kf)).astype(np.float32)
    w1[3] = 0
    b1 = (rng.standard_normal(n1) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((n1, n2)) / np.sqrt(n1)).astype(np.float32)
    b2 = (rng.standard_normal(n2) * 0.1).astype(np.float32)
    idx = ball_query(1.2, ns, dev(xyz), dev(new_xyz))
    ii = idx.cpu().numpy().astype(np.int64)
    bidx = np.arange(b)[:, None, None]
    rel32 = (xyz[bidx, ii] - new_xyz[:, :, None, :]).astype(np.float64)
    rows = np.concatenate([rel32, np.zeros(rel32.shape[:3] + (1,)), feat[bidx, ii].astype(np.float64)], -1).reshape(-1, 4 + kf)
    h = np.maximum(rows @ w1.astype(np.float64) + b1, 0)
    ref = np.maximum(h @ w2.astype(np.float64) + b2, 0)
    p = linear_rows(dev(feat).reshape(b * n, kf), dev(np.ascontiguousarray(w1[4:]))).view(b, n, n1)
    got = sa_mlp_pair(p, dev(xyz), dev(new_xyz), idx, dev(np.ascontiguousarray(w1[:3])), dev(b1), dev(w2), dev(b2), True, False)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=3e-5, atol=3e-5 * np.abs(ref).max())
    wide = torch.zeros((b, n, n1 + 32), device="cuda")  # P as a column block of a wider matrix
    wide[:, :, 16:16 + n1] = p
    assert torch.equal(sa_mlp_pair(wide[:, :, 16:16 + n1], dev(xyz), dev(new_xyz), idx, dev(np.ascontiguousarray(w1[:3])), dev(b1), dev(w2),
                                   dev(b2), True, False), got)
    gotp = sa_mlp_pair(p, dev(xyz), dev(new_xyz), idx, dev(np.ascontiguousarray(w1[:3])), dev(b1), dev(w2), dev(b2), True, True)
    np.testing.assert_allclose(gotp.cpu().numpy(), ref.reshape(b * m, ns, n2).max(1), rtol=3e-5, atol=3e-5 * np.abs(ref).max())
    x = sa_mlp_layer(dev(feat), dev(w1), dev(b1), True, False, xyz=dev(xyz), new_xyz=dev(new_xyz), idx=idx)
    two = sa_mlp_layer(x, dev(w2), dev(b2), True, True, groups=(b, m, ns))
    torch.testing.assert_close(gotp, two, rtol=3e-5, atol=3e-5 * float(np.abs(ref).max()))


def test_both_scales_in_one_launch_equal_a_launch_per_scale():
    """PointnetSAModuleMSG with two scales of the same widths: v3d_sa_mlp_pair2 (one launch) == v3d_sa_mlp_pair per scale, bit for bit
    (the same kernel body per scale), at RoI-grid pooling's shape and a small one."""
    from gpu_util import randomize_bn
    from vision3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(6)
    for c, mlp, n, m in ((512, [512, 192, 96], 2048, 1600), (16, [16, 32, 32], 5000, 333)):
        sa = PointnetSAModuleMSG(npoint=-1, radii=[0.8, 1.6], nsamples=[16, 32], mlps=[list(mlp), list(mlp)], use_xyz=True)
        randomize_bn(sa, 3)
        sa = sa.cuda().eval()
        xyz = torch.from_numpy(np.stack([synth.make_cloud(6)[:n, :3], synth.make_cloud(7)[:n, :3]])).cuda()
        new_xyz = (xyz[:, :m] + 0.1).contiguous()
        feat = torch.randn(2, n, c, device="cuda")
        with torch.no_grad():
            both = sa.fused_forward(xyz, feat, new_xyz).clone()
            sa.PAIR_BOTH_SCALES = False
            each = sa.fused_forward(xyz, feat, new_xyz)
        assert torch.equal(both, each) and both.shape == (2, m, 2 * mlp[-1])


def test_fused_keypoint_features_equal_the_op_by_op_path():
    """PV_RCNN.point_feature_extract in inference writes every set-abstraction scale and the BEV lookup into ONE point-major matrix
    (sa_mlp `ldo`, v3d_bev_gather_keypoints) and RoI-grid pooling reads point-major rows with a permuted first reduction layer:
    keypoint features bit-identical to the cat / transpose path (same kernels, same summation order; the BEV grid by the same fp32
    statements), pooled features equal up to the summation order of the first reduction layer."""
    from gpu_util import randomize_bn
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = PV_RCNN(cfg)
    randomize_bn(model, 2)
    model = model.cuda().eval()
    clouds = [synth.make_cloud(0), synth.make_cloud(1)]
    item = Preprocessor(cfg, seed=0)(dict(points=clouds, anchors=AnchorGenerator(cfg).anchors.cuda()))
    with torch.no_grad():
        model.cnn.pad_generator = torch.Generator(device="cuda").manual_seed(3)
        item = model.proposal(item)
        kp = item["keypoints"]
        kp[0, 5] = 1e4  # keypoints outside the map: the clamp of the grid coordinates
        kp[1, 7, :2] = -50.0
        fused = model.point_feature_extract(item, item["_cnn_features"], item["_bev_map"])
        assert fused.shape == (2, 512, 2048) and fused.transpose(1, 2).is_contiguous()
        xyz, refl = item["points"].split([3, 1], dim=-1)
        pooled = model._pointnets([(xyz, refl), *item["_cnn_features"]], kp)
        plain = torch.cat([*pooled, model.bev.forward_torch(item["_bev_map"], kp)], dim=1)
        assert torch.equal(fused, plain)
        assert torch.equal(model.bev(item["_bev_map"], kp), model.bev.forward_torch(item["_bev_map"], kp))
        props = torch.from_numpy(np.stack([synth.make_gt_boxes(0)[:20], synth.make_gt_boxes(1)[:20]])).cuda()
        samples = torch.rand((2, 20, cfg.GRIDPOOL.NUM_GRIDPOINTS, 3), generator=torch.Generator().manual_seed(5)).cuda()
        got = model.roi_grid_pool(props, kp, fused, samples)
        pts = model.roi_grid_pool.sample_gridpoints(props, samples)
        assert torch.equal(pts, model.roi_grid_pool.sample_gridpoints_torch(props, samples))  # v3d_roi_grid_points == the torch statements
        _, chan_major = model.roi_grid_pool.pnet(kp, plain.contiguous(), pts.reshape(2, -1, 3).contiguous())
        per_box = chan_major.reshape(2, -1, 20, pts.shape[2]).permute(0, 2, 1, 3).reshape(2, 20, -1)
        ref = torch.nn.Sequential.forward(model.roi_grid_pool.reduction.double(), per_box.double())
        model.roi_grid_pool.reduction.float()
        torch.testing.assert_close(got.double(), ref, rtol=2e-5, atol=2e-5 * ref.abs().max().item())


def test_roi_grid_points_trig_equals_torch():
    """v3d_roi_grid_points with the yaw's cos / sin computed inside the launch (cosf / sinf of the device library) against torch.cos /
    torch.sin: bit-identical over a million yaws, large arguments and the special values -- what lets RoiGridPool.sample_gridpoints
    drop its two trig launches; and the module's grid points equal the op-by-op statements."""
    import math
    from vision3d_amd import _lib as L
    g = torch.Generator().manual_seed(0)
    yaw = torch.cat([(torch.rand(1 << 20, generator=g) * 2 - 1) * math.pi, (torch.rand(1 << 18, generator=g) * 2 - 1) * 100.0,
                     torch.tensor([0.0, -0.0, math.pi, -math.pi, math.pi / 2, 1e-8, 1e6, -3e4, float("inf"), float("nan")])]).cuda()
    nb = yaw.numel()
    boxes = torch.zeros(nb, 7, device="cuda")
    boxes[:, 3] = 1
    boxes[:, 4] = 1
    boxes[:, 6] = yaw
    # local offsets (1, 0, 0) and (0, 1, 0): the points are then (cos, sin, 0) and (-sin, cos, 0) exactly
    smp = torch.tensor([[1.5, 0.5, 0.5], [0.5, 1.5, 0.5]], device="cuda").expand(nb, 2, 3).contiguous()
    out = torch.empty(nb, 2, 3, device="cuda")
    L.check(L.lib().v3d_roi_grid_points(L.ptr(boxes), L.ptr(smp), None, None, nb, 2, L.ptr(out), L.stream_ptr()), "roi_grid_points")
    c, s = torch.cos(yaw), torch.sin(yaw)
    for got, ref in ((out[:, 0, 0], c), (out[:, 0, 1], s), (out[:, 1, 0], 0 - s), (out[:, 1, 1], c)):
        assert ((got == ref) | (got.isnan() & ref.isnan())).all()
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector.roi_grid_pool import RoiGridPool
    pool = RoiGridPool(second_car_cfg()).cuda().eval()
    props = torch.from_numpy(np.stack([synth.make_gt_boxes(0)[:20], synth.make_gt_boxes(1)[:20]])).cuda()
    samples = torch.rand((2, 20, 16, 3), generator=g).cuda()
    with torch.no_grad():
        pts = pool.sample_gridpoints(props, samples)
        assert torch.equal(pts, pool.sample_gridpoints_torch(props, samples))
        pool.TORCH_TRIG = True
        assert torch.equal(pts, pool.sample_gridpoints(props, samples))


def test_voxel_centers_equal_the_torch_statements():
    """SparseCNNBase.to_global on v3d_voxel_centers == flip / float / multiply / add op by op, every stride of the config."""
    from vision3d_amd import spconv
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import sparse_cnn
    cfg = second_car_cfg()
    cnn = sparse_cnn.CNN_FACTORY[cfg.CNN](cfg).cuda()
    g = torch.Generator().manual_seed(0)
    ind = torch.stack([torch.zeros(5000, dtype=torch.int32), torch.randint(0, 41, (5000,), generator=g, dtype=torch.int32),
                       torch.randint(0, 1600, (5000,), generator=g, dtype=torch.int32),
                       torch.randint(0, 1408, (5000,), generator=g, dtype=torch.int32)], dim=1).cuda()
    vol = spconv.SparseConvTensor(torch.randn(5000, 16, device="cuda"), ind, cnn.grid_shape, 1)
    for stride in cfg.STRIDES:
        xyz, feat = cnn.to_global(stride, vol)
        ref_xyz, ref_feat = cnn.to_global_torch(stride, vol)
        assert torch.equal(xyz, ref_xyz) and torch.equal(feat, ref_feat)


def test_bev_bilinear_equals_grid_sample():
    """v3d_bev_bilinear == F.grid_sample(bilinear, zeros, align_corners=True) on a (B, 1, K, 2) grid, including points on and
    beyond the border (the gatherer clamps, the kernel must still zero-pad like torch)."""
    import torch.nn.functional as F
    from vision3d_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    fmap = torch.randn(2, 37, 25, 19, generator=g).cuda()
    grid = (torch.rand(2, 1, 400, 2, generator=g) * 2.4 - 1.2).cuda()
    grid[0, 0, 0] = torch.tensor([-1.0, -1.0])
    grid[0, 0, 1] = torch.tensor([1.0, 1.0])
    grid[1, 0, 2] = torch.tensor([0.0, 1.0])
    ref = F.grid_sample(fmap, grid, align_corners=True).squeeze(2)
    out = torch.empty_like(ref)
    L.check(L.lib().v3d_bev_bilinear(L.ptr(fmap), L.ptr(grid.reshape(2, 400, 2).contiguous()), 2, 37, 25, 19, 400, L.ptr(out),
                                     L.stream_ptr()), "bev_bilinear")
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    # and through the module (eval: the custom lookup; with autograd: torch's)
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector.layers import BEVFeatureGatherer
    cfg = second_car_cfg()
    gat = BEVFeatureGatherer(cfg, torch.tensor(cfg.GRID_BOUNDS[:3]), torch.tensor(cfg.VOXEL_SIZE)).cuda()
    bev = torch.randn(1, 16, 200, 176, generator=g).cuda()
    kp = (torch.rand(1, 300, 3, generator=g) * torch.tensor([80.0, 90.0, 4.0]) + torch.tensor([-5.0, -45.0, -3.0])).cuda()
    with torch.no_grad():
        fast = gat(bev, kp)
    slow = gat(bev.requires_grad_(True), kp)
    torch.testing.assert_close(fast, slow.detach(), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_pv_rcnn_two_frames_in_flight_equal_frame_by_frame_inference():
    """PV_RCNN.inference_begin / _end / _collect (stage 1 of the next frame queued before this frame's row counts are read; frames
    alternate between two plan arenas; every wait on the frame's own event), with the keypoints of all frames sampled by one
    batched launch (prefetch_keypoints_many), return what PV_RCNN.inference returns frame by frame, in order, for a stream of
    different clouds."""
    import torch
    from vision3d_amd import synth
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = PV_RCNN(cfg).cuda().eval()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    pre = Preprocessor(cfg, seed=0)
    clouds = [[torch.from_numpy(synth.make_cloud(s, 16384)).cuda()] for s in range(5)]
    samples = torch.rand(1, cfg.NUM_CLASSES * model.proposal_layer.TOPK, cfg.GRIDPOOL.NUM_GRIDPOINTS, 3, device="cuda")

    def item(i):
        return pre(dict(points=clouds[i], anchors=anchors))
    with torch.no_grad():
        want = [[t.clone() for t in model.inference(item(i), samples)] for i in range(len(clouds))]
        got, prev = [], None
        ahead = model.prefetch_keypoints_many([item(j) for j in range(len(clouds))])  # all samplings in ONE launch (a workgroup per cloud)
        st = model.inference_begin(ahead[0], 0)
        for i in range(len(clouds)):
            nxt = model.inference_begin(ahead[i + 1], (i + 1) % 2) if i + 1 < len(clouds) else None
            h = model.inference_end(st, samples)
            if prev is not None:
                got.append([t.clone() for t in model.inference_collect(prev)])
            prev, st = h, nxt
        got.append([t.clone() for t in model.inference_collect(prev)])
    assert len(got) == len(want)
    assert len({w[0].shape[0] for w in want}) >= 1
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert torch.equal(a, b)
