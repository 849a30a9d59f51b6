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
