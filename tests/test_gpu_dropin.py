"""GPU: the reference's own entry points reach the native path.

The `sys.modules` aliasing of INTEGRATION.md is applied, then the body of vision3d/inference.py:20-38 is run as restated
logic against the `vision3d.*` NAMES (no reference file is imported or shipped): cfg -> AnchorGenerator -> Preprocessor ->
Second(cfg).cuda().eval() -> preprocessor(dict(points=[points], anchors=anchors)) -> .cuda() on the item -> model.inference(item).
That call must run backbone plan -> csrc/dense_conv.hip -> csrc/proposal.hip: torch's convolution is disabled while it runs.
"""
import importlib
import sys

import numpy as np
import pytest
import torch

from gpu_util import assert_features_close, assert_fp32_class, randomize_bn
from vision3d_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def aliased():
    import vision3d_amd
    import vision3d_amd._C, vision3d_amd.core, vision3d_amd.detector, vision3d_amd.ops
    import vision3d_amd.pointnet2, vision3d_amd.spconv
    names = {
        "vision3d": vision3d_amd, "vision3d.ops": vision3d_amd.ops, "vision3d._C": vision3d_amd._C,
        "vision3d.core": vision3d_amd.core, "vision3d.detector": vision3d_amd.detector,
        "spconv": vision3d_amd.spconv, "pointnet2": vision3d_amd.pointnet2,
        "pointnet2.pointnet2_utils": vision3d_amd.pointnet2.pointnet2_utils,
        "pointnet2.pointnet2_modules": vision3d_amd.pointnet2.pointnet2_modules,
    }
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules.update(names)
    yield
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


class _NoTorchConv:
    """While active, any torch convolution raises: the call under test must not fall back to MIOpen."""

    def __enter__(self):
        self.saved = (torch.nn.functional.conv2d, torch.conv2d, torch.nn.Conv2d.forward)

        def boom(*a, **k):
            raise AssertionError("torch convolution reached: the reference surface did not take the native path")
        torch.nn.functional.conv2d = boom
        torch.conv2d = boom
        torch.nn.Conv2d.forward = boom
        return self

    def __exit__(self, *exc):
        torch.nn.functional.conv2d, torch.conv2d, torch.nn.Conv2d.forward = self.saved


def test_reference_inference_script_body_runs_native(aliased):
    from vision3d.core import AnchorGenerator, Preprocessor      # inference.py:5
    from vision3d.core.config import SECOND_CAR, _defaults
    from vision3d.detector import Second                          # inference.py:7
    from vision3d.ops import box_iou_rotated, nms_rotated         # the ops surface resolves too (ops/__init__.py:1-4)
    assert box_iou_rotated is not None and nms_rotated is not None
    cfg = _defaults()
    cfg.merge_from_dict(dict(SECOND_CAR))                         # inference.py:21 merges configs/second/car.yaml
    anchors = AnchorGenerator(cfg).anchors                        # :22
    preprocessor = Preprocessor(cfg)                              # :23
    torch.manual_seed(0)
    model = Second(cfg)
    randomize_bn(model, 3)
    model = model.cuda().eval()                                   # :24 (no checkpoint exists: random weights)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(sd, strict=True)                        # :26
    points = synth.make_cloud(11)                                 # :32-33 read a velodyne .bin: (N, 4) float32
    with torch.no_grad():                                         # :34
        item = preprocessor(dict(points=[points], anchors=anchors))  # :35
        for key in ["points", "features", "coordinates", "occupancy", "anchors"]:
            item[key] = item[key].cuda()                          # :36-37
        with _NoTorchConv():
            boxes, batch_idx, class_idx, scores = model.inference(item)  # :38
        # the same frame through the raw-points entry (voxelizer inside the plan): identical kernels downstream
        ref = model.inference_points([torch.from_numpy(points).cuda()], anchors.cuda())
    assert boxes.shape[1] == 7 and len(boxes) == len(scores) > 0
    for got, want in zip((boxes, batch_idx, class_idx, scores), ref):
        np.testing.assert_array_equal(got.cpu().numpy(), want.cpu().numpy())


def test_eval_forward_item_is_native_and_matches_module_path(aliased):
    """Second.forward(item) in eval mode (what an evaluation loop around train.py:63's call would run): P_cls / P_reg from the
    native path against the module-by-module path (torch convolutions, taken when autograd is on), feature tolerance."""
    from vision3d.core import AnchorGenerator, Preprocessor
    from vision3d.core.config import second_car_cfg
    from vision3d.detector import Second
    cfg = second_car_cfg()
    torch.manual_seed(1)
    model = Second(cfg)
    randomize_bn(model, 4)
    model = model.cuda().eval()
    clouds = [synth.make_cloud(12), synth.make_cloud(13)[:15000]]
    with torch.no_grad():
        item = Preprocessor(cfg, seed=0)(dict(points=clouds, anchors=AnchorGenerator(cfg).anchors.cuda()))
        with _NoTorchConv():
            out = model(dict(item))
        p_cls, p_reg = out["P_cls"].clone(), out["P_reg"].clone()
    ref = model(dict(item))  # autograd on -> module path (MIOpen fp32)
    assert p_cls.shape == ref["P_cls"].shape == (2, 1, 2, 200, 176) and p_reg.shape == ref["P_reg"].shape == (2, 1, 2, 200, 176, 7)
    # elementwise against the module path (torch fp32), strict bar against float64 from the definition (oracle/second_cpu.py)
    from gpu_util import numpy_state_dict
    from oracle import second_cpu
    ref64 = second_cpu.second_forward64(numpy_state_dict(model), clouds, cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS)
    assert_fp32_class(p_cls.cpu().numpy(), ref["P_cls"].detach().cpu().numpy(), "P_cls native vs module path",
                      ref64["cls"].reshape(p_cls.shape))
    assert_fp32_class(p_reg.cpu().numpy(), ref["P_reg"].detach().cpu().numpy(), "P_reg native vs module path",
                      ref64["reg"].reshape(p_reg.shape[0], p_reg.shape[1], 7, *p_reg.shape[2:5]).transpose(0, 1, 3, 4, 5, 2))
