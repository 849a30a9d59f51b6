"""GPU: fused target assignment (csrc/targets.hip) vs the op-by-op statement of the reference
(core/proposal_targets.py: IoU matrix + torch Matcher + encode) on the same device -- labels and masks exact,
regression targets to 1e-5 -- for one and three classes, with and without the low-quality rule, and without any
ground truth.  (tests/test_gpu_second.py pins the same entry point against targets captured from the reference.)"""
import numpy as np
import pytest
import torch

from vision3d_amd import synth
from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


def three_class_cfg():
    cfg = second_car_cfg().clone()
    cfg.ANCHORS = [dict(cfg.ANCHORS[0]),
                   dict(cfg.ANCHORS[0], wlh=[0.6, 0.8, 1.73], center_z=-0.6, iou_thresh=[0.20, 0.35]),
                   dict(cfg.ANCHORS[0], wlh=[0.6, 1.76, 1.73], center_z=-0.6, iou_thresh=[0.20, 0.35])]
    cfg.NUM_CLASSES = 3
    return cfg


def make_item(seed, n_cls):
    gt = synth.make_gt_boxes(seed)
    rng = np.random.default_rng(seed)
    cls = rng.integers(0, n_cls, len(gt))
    if n_cls > 1:  # give the small classes boxes of their own size so that they get positives
        gt = gt.copy()
        gt[cls == 1, 3:6] = [0.6, 0.8, 1.73]
        gt[cls == 2, 3:6] = [0.6, 1.76, 1.73]
    return dict(boxes=torch.from_numpy(gt), class_idx=torch.from_numpy(cls).long(),
                box_ignore=torch.zeros(len(gt), dtype=torch.bool))


@pytest.mark.parametrize("n_cls,low_quality", [(1, False), (1, True), (3, False), (3, True)])
def test_fused_matches_torch_statement(n_cls, low_quality):
    from vision3d_amd.core import ProposalTargetAssigner
    cfg = three_class_cfg() if n_cls == 3 else second_car_cfg().clone()
    cfg.ALLOW_LOW_QUALITY_MATCHES = low_quality
    assigner = ProposalTargetAssigner(cfg)
    for seed in (0, 5):
        fused = assigner(dict(make_item(seed, n_cls)))
        ref = assigner.forward_torch(dict(make_item(seed, n_cls)))
        assert int(fused["M_reg"].sum()) > 0, "test data must produce positives"
        for k in ("G_cls", "M_cls", "M_reg"):
            assert fused[k].dtype == ref[k].dtype and fused[k].shape == ref[k].shape, k
            assert torch.equal(fused[k], ref[k]), k
        np.testing.assert_allclose(fused["G_reg"].cpu().numpy(), ref["G_reg"].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_no_ground_truth_and_absent_class():
    from vision3d_amd.core import ProposalTargetAssigner
    cfg = three_class_cfg()
    assigner = ProposalTargetAssigner(cfg)
    empty = dict(boxes=torch.zeros((0, 7)), class_idx=torch.zeros((0,), dtype=torch.long), box_ignore=torch.zeros((0,), dtype=torch.bool))
    out = assigner(dict(empty))
    assert not out["G_cls"].any() and out["M_cls"].all() and not out["M_reg"].any() and not out["G_reg"].any()
    item = make_item(2, 1)  # every box is class 0: classes 1 and 2 see no ground truth
    fused, ref = assigner(dict(item)), assigner.forward_torch(dict(item))
    for k in ("G_cls", "M_cls", "M_reg"):
        assert torch.equal(fused[k], ref[k]), k
    np.testing.assert_allclose(fused["G_reg"].cpu().numpy(), ref["G_reg"].cpu().numpy(), rtol=1e-5, atol=1e-6)
