"""GPU parity: rotated IoU / NMS through the C ABI vs the reference's golden vectors and oracle/.

Bars: IoU within 1e-4 relative (+1e-6 abs) of the reference -- the kernel follows the reference's
host-branch arithmetic, so the observed agreement is reported as a bit-exact fraction too; NMS keep
lists IDENTICAL (index order included) on fixtures whose evaluated pairs all sit > 1e-5 from the
threshold (margins stored in the fixture)."""
import numpy as np
import pytest
import torch

from gpu_util import dev

pytestmark = pytest.mark.gpu


def gpu_iou(b1, b2):
    from vision3d_amd.ops import box_iou_rotated
    return box_iou_rotated(dev(b1, torch.float32), dev(b2, torch.float32)).cpu().numpy()


def check_iou(got, ref, what):
    exact = float((got == ref).mean())
    print(f"[iou] {what}: bit-exact fraction {exact:.6f}, max abs diff {np.abs(got - ref).max():.3e}")
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6, err_msg=what)
    assert exact > 0.99, what


@pytest.mark.parametrize("tag", ["kat", "deg", "rad", "far", "dense"])
def test_iou_golden(golden_iou, tag):
    check_iou(gpu_iou(golden_iou[f"iou_{tag}_b1"], golden_iou[f"iou_{tag}_b2"]), golden_iou[f"iou_{tag}"], tag)


def test_iou_degenerates(golden_iou):
    b = golden_iou["iou_degen_b"]
    check_iou(gpu_iou(b, b), golden_iou["iou_degen"], "degenerate")


def test_iou_target_assignment_shape(golden_iou):
    """27 gt x 70400 anchors (core/proposal_targets.py:29-34), radians fed as degrees (H1)."""
    from vision3d_amd.core.anchor_generator import AnchorGenerator
    from vision3d_amd.core.config import second_car_cfg
    anchors = AnchorGenerator(second_car_cfg()).anchors.view(-1, 7).numpy()
    bev = [0, 1, 3, 4, 6]
    iou = gpu_iou(golden_iou["ta_gt"][:, bev], anchors[:, bev])
    nz = np.argwhere(iou != 0)
    np.testing.assert_array_equal(nz, golden_iou["ta_iou_nz_idx"])
    check_iou(iou[nz[:, 0], nz[:, 1]], golden_iou["ta_iou_nz_val"], "target-assign nonzeros")


def test_iou_vs_oracle_random_and_edges(oracle):
    rng = np.random.default_rng(21)

    def boxes(n, ang, spread):
        return np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(0.2, 5, (n, 2)),
                               rng.uniform(-1, 1, (n, 1)) * ang], 1).astype(np.float32)
    for ang, spread, m, n in ((180, 10, 300, 517), (3.2, 4, 65, 1000), (180, 1, 1, 1), (90, 2, 33, 257)):
        b1, b2 = boxes(m, ang, spread), boxes(n, ang, spread)
        check_iou(gpu_iou(b1, b2), oracle.box_iou_rotated(b1, b2), f"random {m}x{n}")
    assert gpu_iou(np.zeros((0, 5), np.float32), boxes(7, 1, 1)).shape == (0, 7)
    assert gpu_iou(boxes(7, 1, 1), np.zeros((0, 5), np.float32)).shape == (7, 0)


def test_iou_3d_vs_oracle(oracle):
    """box_iou_rotated_3d (a stub upstream): BEV intersection of the 2-D operator x z overlap / union of the
    volumes, bit-exact against oracle/ (same float32 operation order) on KITTI-like boxes, dense clusters, stacked and
    degenerate boxes; ragged shapes; consistency with the 2-D operator when the z extents coincide."""
    from vision3d_amd.ops import box_iou_rotated, box_iou_rotated_3d
    rng = np.random.default_rng(33)

    def boxes(n, spread, ang):
        return np.concatenate([rng.uniform(-spread, spread, (n, 2)), rng.uniform(-2, 1, (n, 1)), rng.uniform(0.3, 4.5, (n, 3)),
                               rng.uniform(-1, 1, (n, 1)) * ang], 1).astype(np.float32)
    for spread, ang, m, n in ((30, 3.2, 300, 517), (4, 180, 65, 1000), (2, 90, 33, 257), (1, 1, 1, 1)):
        b1, b2 = boxes(m, spread, ang), boxes(n, spread, ang)
        got = box_iou_rotated_3d(dev(b1, torch.float32), dev(b2, torch.float32)).cpu().numpy()
        ref = oracle.box_iou_rotated_3d(b1, b2)
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32), err_msg=f"{m}x{n}")
        assert (got >= 0).all() and (got <= 1 + 1e-6).all() and (m * n < 100 or (got > 0).any())
    deg = boxes(12, 1, 30)
    deg[0, 5] = 0.0            # zero height
    deg[1, 3] = 0.0            # zero width
    deg[2] = deg[3]            # identical boxes
    deg[4, :2] = deg[5, :2]
    deg[4, 2] = deg[5, 2] + deg[5, 5] / 2 + deg[4, 5] / 2  # stacked: z extents touch
    got = box_iou_rotated_3d(dev(deg, torch.float32), dev(deg, torch.float32)).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), oracle.box_iou_rotated_3d(deg, deg).view(np.uint32))
    assert got[0].max() == 0 and got[1].max() == 0 and abs(got[2, 3] - 1) < 1e-6 and got[4, 5] == 0
    same_z = boxes(40, 3, 90)
    same_z[:, 2], same_z[:, 5] = -1.0, 1.5
    g3 = box_iou_rotated_3d(dev(same_z, torch.float32), dev(same_z, torch.float32)).cpu().numpy()
    g2 = box_iou_rotated(dev(same_z[:, [0, 1, 3, 4, 6]].copy(), torch.float32), dev(same_z[:, [0, 1, 3, 4, 6]].copy(), torch.float32)).cpu().numpy()
    np.testing.assert_allclose(g3, g2, rtol=2e-6, atol=1e-7)  # equal heights: the 3-D ratio reduces to the BEV one
    assert box_iou_rotated_3d(dev(np.zeros((0, 7), np.float32), torch.float32), dev(deg, torch.float32)).shape == (0, 12)


def gpu_nms(boxes, scores, thr):
    from vision3d_amd.ops import nms_rotated
    return nms_rotated(dev(boxes, torch.float32), dev(scores, torch.float32), thr).cpu().numpy()


@pytest.mark.parametrize("tag,thr", [("nms100", 0.01), ("nms100", 0.5), ("nms800", 0.01), ("nms800", 0.5),
                                     ("nms4096", 0.01), ("nms4096", 0.5), ("nmsdeg", 0.3)])
def test_nms_golden(golden_iou, tag, thr):
    key = f"{tag}_t{int(thr * 100):02d}"
    assert golden_iou[key + "_margin"] > 1e-5
    keep = gpu_nms(golden_iou[f"{tag}_boxes"], golden_iou[f"{tag}_scores"], thr)
    np.testing.assert_array_equal(keep, golden_iou[key + "_keep"])
    assert keep.dtype == np.int64


@pytest.mark.parametrize("groups", [1, 3, 8])
def test_batched_nms_golden(golden_iou, groups):
    from vision3d_amd.ops import batched_nms_rotated
    keep = batched_nms_rotated(dev(golden_iou[f"bnms{groups}_boxes"]), dev(golden_iou[f"bnms{groups}_scores"]),
                               dev(golden_iou[f"bnms{groups}_idxs"]), 0.01).cpu().numpy()
    np.testing.assert_array_equal(keep, golden_iou[f"bnms{groups}_keep"])


def test_nms_edges_and_properties(oracle):
    from vision3d_amd.ops import batched_nms_rotated, nms_rotated
    e = nms_rotated(torch.zeros(0, 5).cuda(), torch.zeros(0).cuda(), 0.5)
    assert e.shape == (0,) and e.dtype == torch.int64 and e.is_cuda
    assert batched_nms_rotated(torch.zeros(0, 5).cuda(), torch.zeros(0).cuda(), torch.zeros(0).long().cuda(), 0.5).shape == (0,)
    one = gpu_nms(np.array([[1, 2, 3, 4, 5]], np.float32), np.array([0.3], np.float32), 0.5)
    np.testing.assert_array_equal(one, [0])
    # identical boxes: only the best survives; equal scores resolve to the lower index
    same = np.tile(np.array([[5, 5, 4, 2, 30]], np.float32), (70, 1))
    np.testing.assert_array_equal(gpu_nms(same, np.arange(70, dtype=np.float32), 0.5), [69])
    np.testing.assert_array_equal(gpu_nms(same, np.ones(70, np.float32), 0.5), [0])
    # large N (non power of two, multi-launch sort path): against the oracle + idempotence + sortedness
    rng = np.random.default_rng(5)
    n = 9000
    boxes = np.concatenate([rng.uniform(0, 150, (n, 2)), rng.normal([1.6, 3.9], 0.2, (n, 2)),
                            rng.uniform(-3.2, 3.2, (n, 1))], 1).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32)
    keep = gpu_nms(boxes, scores, 0.1)
    if oracle.nms_margin(boxes, scores, 0.1) > 1e-5:
        np.testing.assert_array_equal(keep, oracle.nms_rotated(boxes, scores, 0.1))
    assert np.all(np.diff(scores[keep]) < 0)
    again = gpu_nms(boxes[keep], scores[keep], 0.1)
    np.testing.assert_array_equal(again, np.arange(len(keep)))


def test_version_probes():
    from vision3d_amd import _C
    assert _C.get_cuda_version().startswith("HIP")
    assert "clang" in _C.get_compiler_version()


def test_nms_tie_at_threshold_follows_the_chosen_rule():
    """IoU == threshold exactly: the CPU rule (>=, the parity target) suppresses, the CUDA rule (>) keeps."""
    from vision3d_amd.ops import box_iou_rotated, nms_rotated
    boxes = torch.tensor([[0.0, 0.0, 2.0, 2.0, 0.0], [1.0, 0.0, 2.0, 2.0, 0.0], [10.0, 10.0, 2.0, 2.0, 0.0]]).cuda()
    scores = torch.tensor([0.9, 0.8, 0.7]).cuda()
    thr = float(box_iou_rotated(boxes[:1], boxes[1:2]).item())  # the exact float the kernels compare
    assert abs(thr - 1.0 / 3.0) < 1e-6
    assert nms_rotated(boxes, scores, thr).tolist() == [0, 2]
    assert nms_rotated(boxes, scores, thr, rule="cuda").tolist() == [0, 1, 2]
    lower = float(np.nextafter(np.float32(thr), np.float32(0)))
    assert nms_rotated(boxes, scores, lower, rule="cuda").tolist() == [0, 2]
    # thr = 0: ">=" suppresses even disjoint boxes (0 >= 0), ">" does not
    assert nms_rotated(boxes, scores, 0.0).tolist() == [0]
    assert nms_rotated(boxes, scores, 0.0, rule="cuda").tolist() == [0, 2]
