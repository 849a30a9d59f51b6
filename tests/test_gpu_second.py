"""GPU parity: SECOND forward end to end (voxelize -> VFE -> 14 sparse layers -> .dense() -> RPN -> heads)
vs the CPU restatement (oracle/second_cpu.py), features within 1e-4 relative; proposal post-processing
and target assignment vs vectors captured from the reference itself (tests/golden/core.npz)."""
import numpy as np
import pytest
import torch

from gpu_util import assert_features_close, assert_fp32_class, dev, numpy_state_dict, randomize_bn
from vision3d_amd import synth
from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


def build_model(seed=0):
    from vision3d_amd.detector import Second
    torch.manual_seed(seed)
    model = Second(second_car_cfg())
    randomize_bn(model, seed)
    with torch.no_grad():
        model.head.conv_cls.weight.normal_(0, 0.05)
        model.head.conv_reg.weight.normal_(0, 0.02)
    return model.cuda().eval()


@pytest.mark.parametrize("seeds", [[0], [1, 2]])
def test_second_forward_matches_cpu_restatement(seeds):
    from oracle import second_cpu
    from vision3d_amd.core import Preprocessor
    cfg = second_car_cfg()
    model = build_model(0)
    clouds = [synth.make_cloud(s) for s in seeds]
    ref = second_cpu.second_forward(numpy_state_dict(model), clouds, cfg.VOXEL_SIZE, cfg.GRID_BOUNDS,
                                    cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS)
    ref64 = second_cpu.second_forward64(numpy_state_dict(model), clouds, cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS)
    item = Preprocessor(cfg)(dict(points=[c.copy() for c in clouds]))
    np.testing.assert_array_equal(item["coordinates"].cpu().numpy(), ref["coords"])
    np.testing.assert_array_equal(item["occupancy"].cpu().numpy(), ref["occupancy"])
    np.testing.assert_array_equal(item["features"].cpu().numpy(), ref["voxels"])
    assert item["points"].shape == (len(seeds), 16384, 4)
    with torch.no_grad():
        bev = model.cnn(item["voxel_mean"], item["coordinates"], item["batch_size"])
        assert bev.shape == (len(seeds), 128, 200, 176)
        assert_fp32_class(bev.cpu().numpy(), ref["bev"], "BEV map after sparse backbone", ref64["bev"])
        # active BEV cells identical (index work is exact)
        np.testing.assert_array_equal((bev.abs().sum(1) > 0).cpu().numpy(), np.abs(ref["bev"]).sum(1) > 0)
        rpn = model.rpn(bev)
        # (the 128-channel map after 14 sparse + 7 dense layers, measured on MI355X in round 6: 2.9e-4 ... 4.9e-4 strict against float64
        #  where torch's CPU fp32 modules show 1.3e-4 ... 2.0e-4 -- entries just above the 1e-3 cut; 3.8e-5 vs 2.9e-5 above 1e-2 of
        #  the maximum.  The model's OUTPUTS below hold the plain 2e-4 bar: P_cls 8e-8, P_reg 0.7e-4 ... 1.2e-4.)
        assert_fp32_class(rpn.cpu().numpy(), ref["rpn"], "RPN output", ref64["rpn"], own_factor=3.0)
        out = model(item)
    b = len(seeds)
    assert out["P_cls"].shape == (b, 1, 2, 200, 176) and out["P_reg"].shape == (b, 1, 2, 200, 176, 7)
    assert_fp32_class(out["P_cls"].cpu().numpy().reshape(ref["cls"].shape), ref["cls"], "P_cls", ref64["cls"])
    reg = out["P_reg"].permute(0, 1, 5, 2, 3, 4).reshape(ref["reg"].shape)
    assert_fp32_class(reg.cpu().numpy(), ref["reg"], "P_reg", ref64["reg"])
    # the un-fused drop-in path (features/occupancy through the VFE module) gives the same BEV map
    item2 = {k: v for k, v in item.items() if k != "voxel_mean"}
    with torch.no_grad():
        f2 = model.vfe(item2["features"], item2["occupancy"])
    # torch's own reduction may associate the 5-slot sum differently: 1 ulp, not exact
    np.testing.assert_allclose(f2.cpu().numpy(), item["voxel_mean"].cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_proposal_layer_inference_matches_reference_golden(golden_core):
    """detector/proposal.py:72-80 on the GPU (top-k, decode, device NMS) vs the reference's own output."""
    from test_host_golden import _load_flat
    from vision3d_amd.core.anchor_generator import AnchorGenerator
    from vision3d_amd.detector.proposal import ProposalLayer
    cfg = second_car_cfg()
    layer = ProposalLayer(cfg)
    _load_flat(layer, golden_core["pl_state"])
    layer = layer.cuda().eval()
    small = second_car_cfg()
    small.GRID_BOUNDS = [0, -8.0, -3, 12.8, 8.0, 1]
    anchors = AnchorGenerator(small).anchors
    assert list(anchors.shape) == list(golden_core["pl_anchor_shape"])
    with torch.no_grad():
        boxes, bidx, cidx, scores = layer.inference(dev(golden_core["pl_fm"]), anchors.cuda())
    assert boxes.is_cuda and bidx.is_cuda
    np.testing.assert_array_equal(bidx.cpu().numpy(), golden_core["pl_bidx"])
    np.testing.assert_array_equal(cidx.cpu().numpy(), golden_core["pl_cidx"])
    np.testing.assert_allclose(scores.cpu().numpy(), golden_core["pl_scores"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(boxes.cpu().numpy(), golden_core["pl_boxes"], rtol=1e-4, atol=1e-4)


def test_second_inference_end_to_end():
    """Second.inference on a full-size frame; post-processing re-derived on the CPU from the GPU's own
    score/box maps (isolates top-k/decode/NMS from feature rounding)."""
    from oracle import second_cpu
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    cfg = second_car_cfg()
    model = build_model(1)
    anchors = AnchorGenerator(cfg).anchors
    item = Preprocessor(cfg)(dict(points=[synth.make_cloud(4), synth.make_cloud(5)], anchors=anchors.cuda()))
    with torch.no_grad():
        # rescale the class head so logits have unit spread: saturated sigmoids (exact fp32 ties among
        # the top-k) would make the CPU/GPU top-k order, hence the NMS outcome, ambiguous
        logits = model.head.conv_cls(model.feature_extract(item))
        model.head.conv_cls.weight.mul_(1.0 / logits.std())
        model.head.conv_cls.bias.zero_()
        boxes, bidx, cidx, scores = model.inference(item)
        cls_map, reg_map = model.head(model.feature_extract(item))
    cls = cls_map.reshape(2, 2, 200, 176).cpu().numpy()
    reg = reg_map.permute(0, 1, 5, 2, 3, 4).reshape(2, 14, 200, 176).cpu().numpy()
    rb, rbi, rci, rs = second_cpu.proposals(cls, reg, anchors.numpy(), 1, 2, 7, cfg.PROPOSAL.TOPK,
                                            [a["score_thresh"] for a in cfg.ANCHORS])
    assert len(boxes) > 0 and boxes.shape[1] == 7
    assert np.all(np.diff(scores.cpu().numpy()) <= 0)
    # NMS invariants of the device result (hold regardless of tie order): per frame every kept pair has
    # IoU < 0.01 and all scores exceed the class threshold
    from oracle import oracle as O
    bnp, binp = boxes.cpu().numpy(), bidx.cpu().numpy()
    for f in range(2):
        sel = bnp[binp == f][:, [0, 1, 3, 4, 6]]
        iou = O.box_iou_rotated(sel, sel)
        np.fill_diagonal(iou, 0)
        assert (iou < 0.01).all()
    assert (scores > 0.3).all()
    cpu_scores_all = torch.from_numpy(cls).sigmoid().reshape(2, -1).topk(cfg.PROPOSAL.TOPK, -1).values.numpy()
    ties = any(len(np.unique(row)) < len(row) for row in cpu_scores_all)
    if not ties:  # without exact score ties the order is unambiguous -> identical detections
        assert len(rb) == len(boxes)
        np.testing.assert_array_equal(bidx.cpu().numpy(), rbi)
        np.testing.assert_allclose(scores.cpu().numpy(), rs, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(boxes.cpu().numpy(), rb, rtol=1e-4, atol=1e-4)
    else:
        print(f"[second inference] exact score ties among the top-k (empty BEV regions share one logit): "
              f"order ambiguous, compared invariants only; GPU {len(boxes)} vs CPU {len(rb)} detections")


def test_target_assigner_matches_reference_golden(golden_core):
    from vision3d_amd.core import ProposalTargetAssigner
    cfg = second_car_cfg()
    gt = torch.from_numpy(synth.make_gt_boxes(0))
    item = dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long), box_ignore=torch.zeros(len(gt), dtype=torch.bool))
    ProposalTargetAssigner(cfg)(item)
    assert item["G_cls"].shape == (1, 2, 200, 176) and item["G_reg"].shape == (1, 2, 200, 176, 7)
    np.testing.assert_array_equal(item["G_cls"].nonzero().cpu().numpy(), golden_core["pta_G_cls_idx"])
    np.testing.assert_array_equal((~item["M_cls"]).nonzero().cpu().numpy(), golden_core["pta_M_cls_zero_idx"])
    mr = item["M_reg"].squeeze(-1)
    np.testing.assert_array_equal(mr.nonzero().cpu().numpy(), golden_core["pta_M_reg_idx"])
    np.testing.assert_allclose(item["G_reg"][mr].cpu().numpy(), golden_core["pta_G_reg_vals"], rtol=1e-5, atol=1e-6)
