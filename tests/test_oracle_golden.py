"""CPU: pin oracle/ (the C restatement) against the reference's golden vectors and oracle/_ref.

Bar: IoU bit-exact vs the reference-built core where it is available, <= 1e-6 abs vs the golden
npz (captured from vision3d._C, whose arithmetic is the same header); NMS keep lists identical.
"""
import hashlib

import numpy as np
import pytest

IOU_TAGS = ["kat", "deg", "rad", "far", "dense"]


@pytest.mark.parametrize("tag", IOU_TAGS)
def test_iou_matches_golden(oracle, golden_iou, tag):
    got = oracle.box_iou_rotated(golden_iou[f"iou_{tag}_b1"], golden_iou[f"iou_{tag}_b2"])
    np.testing.assert_array_equal(got, golden_iou[f"iou_{tag}"])


def test_iou_degenerates_match_golden(oracle, golden_iou):
    b = golden_iou["iou_degen_b"]
    np.testing.assert_array_equal(oracle.box_iou_rotated(b, b), golden_iou["iou_degen"])


def test_iou_known_answers(oracle):
    b = np.array([[0, 0, 2, 2, 0]], np.float32)
    assert oracle.box_iou_rotated(b, b)[0, 0] == 1.0
    assert abs(oracle.box_iou_rotated(b, [[1, 1, 2, 2, 0]])[0, 0] - 1 / 7) < 1e-7
    assert abs(oracle.box_iou_rotated(b, [[0, 0, 2, 2, 45]])[0, 0] - 0.70710678) < 1e-7


def test_iou_target_assign_shape_matches_golden(oracle, golden_iou):
    """27 gt x 70400 anchors, yaw in radians fed to the degrees kernel."""
    from vision3d_amd.core.anchor_generator import AnchorGenerator
    from vision3d_amd.core.config import second_car_cfg
    anchors = AnchorGenerator(second_car_cfg()).anchors.view(-1, 7).numpy()
    bev = [0, 1, 3, 4, 6]
    iou = oracle.box_iou_rotated(golden_iou["ta_gt"][:, bev], anchors[:, bev])
    nz = np.argwhere(iou != 0)
    np.testing.assert_array_equal(nz, golden_iou["ta_iou_nz_idx"])
    np.testing.assert_array_equal(iou[nz[:, 0], nz[:, 1]], golden_iou["ta_iou_nz_val"])


@pytest.mark.parametrize("tag,thr", [("nms100", 0.01), ("nms100", 0.5), ("nms800", 0.01), ("nms800", 0.5),
                                     ("nms4096", 0.01), ("nms4096", 0.5), ("nmsdeg", 0.3)])
def test_nms_matches_golden(oracle, golden_iou, tag, thr):
    key = f"{tag}_t{int(thr * 100):02d}"
    keep = oracle.nms_rotated(golden_iou[f"{tag}_boxes"], golden_iou[f"{tag}_scores"], thr)
    np.testing.assert_array_equal(keep, golden_iou[key + "_keep"])
    assert golden_iou[key + "_margin"] > 1e-5  # keep-set comparison is well posed (H3)


@pytest.mark.parametrize("groups", [1, 3, 8])
def test_batched_nms_matches_golden(oracle, golden_iou, groups):
    keep = oracle.batched_nms_rotated(golden_iou[f"bnms{groups}_boxes"], golden_iou[f"bnms{groups}_scores"],
                                      golden_iou[f"bnms{groups}_idxs"], 0.01)
    np.testing.assert_array_equal(keep, golden_iou[f"bnms{groups}_keep"])


def test_oracle_bit_exact_vs_reference_build(oracle):
    """oracle/_ref is the reference header compiled in place; present wherever it was built."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(5)
    for ang, spread in ((180.0, 10.0), (3.1416, 10.0), (180.0, 3.0), (90.0, 1.0)):
        def boxes(n):
            xy = rng.uniform(-spread, spread, (n, 2))
            wh = rng.uniform(0.2, 5, (n, 2))
            a = rng.uniform(-1, 1, (n, 1)) * ang
            return np.concatenate([xy, wh, a], 1).astype(np.float32)
        b1, b2 = boxes(200), boxes(300)
        np.testing.assert_array_equal(oracle.box_iou_rotated(b1, b2), oracle.box_iou_rotated(b1, b2, use_ref=True))
    # near-duplicates exercise the >16-point introsort branch of the hull sort
    b1 = boxes(300)
    b2 = b1.copy()
    b2[:, :2] += rng.normal(0, 1e-4, (300, 2)).astype(np.float32)
    b2[:, 4] += rng.normal(0, 1e-3, 300).astype(np.float32)
    np.testing.assert_array_equal(oracle.box_iou_rotated(b1, b2), oracle.box_iou_rotated(b1, b2, use_ref=True))
    np.testing.assert_array_equal(oracle.box_iou_rotated(b1, b1), oracle.box_iou_rotated(b1, b1, use_ref=True))


def test_points_in_boxes_matches_golden(oracle, golden_geom):
    from vision3d_amd import synth
    cloud = synth.make_cloud(0)
    assert np.array_equal(np.frombuffer(hashlib.sha256(cloud.tobytes()).digest(), np.uint8), golden_geom["cloud_sha256"])
    shape = tuple(golden_geom["mask_shape"])
    n = shape[0] * shape[1]
    m3 = np.unpackbits(golden_geom["mask3d_packed"])[:n].reshape(shape).astype(bool)
    m2 = np.unpackbits(golden_geom["mask2d_packed"])[:n].reshape(shape).astype(bool)
    assert m3.sum() > 500
    np.testing.assert_array_equal(oracle.points_in_boxes(cloud, golden_geom["boxes"], True), m3)
    np.testing.assert_array_equal(oracle.points_in_boxes(cloud, golden_geom["boxes"], False), m2)


def test_vfe_matches_golden(oracle, golden_core):
    np.testing.assert_allclose(oracle.vfe_mean(golden_core["vfe_feat"], golden_core["vfe_occ"]), golden_core["vfe_out"],
                               rtol=1e-6, atol=1e-7)
