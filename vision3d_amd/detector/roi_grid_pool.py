"""RoI-grid pooling (interface of vision3d/detector/roi_grid_pool.py:10-72): sample grid points inside
each proposal, set-abstract keypoint features around them, flatten, reduce.

The reference draws the grid points with an unseeded torch.rand (roi_grid_pool.py:59);
here `generator` / `samples` make that draw injectable so results are reproducible.
"""
from copy import deepcopy

import torch
from torch import nn

from ..pointnet2.pointnet2_modules import PointnetSAModuleMSG
from .layers import MLP


class RoiGridPool(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.pnet = self.build_pointnet(cfg)
        self.reduction = MLP(cfg.GRIDPOOL.MLPS_REDUCTION)
        self.generator = None

    def build_pointnet(self, cfg):
        return PointnetSAModuleMSG(npoint=-1, radii=cfg.GRIDPOOL.RADII_PN, nsamples=cfg.SAMPLES_PN,
                                   mlps=deepcopy(cfg.GRIDPOOL.MLPS_PN), use_xyz=True)

    @staticmethod
    def rotate_z(points, theta):
        """points (b, n, m, 3) rotated by theta (b, n) about z."""
        c, s = torch.cos(theta)[..., None], torch.sin(theta)[..., None]
        x, y, z = points.unbind(-1)
        return torch.stack((c * x - s * y, s * x + c * y, z), dim=-1)

    def sample_gridpoints(self, boxes, samples=None):
        """boxes (b, n, 7) -> (b, n, m, 3): uniform in the box frame, rotated by yaw, translated."""
        b, n, _ = boxes.shape
        m = self.cfg.GRIDPOOL.NUM_GRIDPOINTS
        if samples is None:
            samples = torch.rand((b, n, m, 3), device=boxes.device, generator=self.generator)
        local = boxes[:, :, None, 3:6] * (samples - 0.5)
        return boxes[:, :, None, 0:3] + self.rotate_z(local, boxes[..., -1])

    def forward(self, proposals, keypoint_xyz, keypoint_features, samples=None):
        b, n, _ = proposals.shape
        m = self.cfg.GRIDPOOL.NUM_GRIDPOINTS
        grid = self.sample_gridpoints(proposals, samples).view(b, -1, 3).contiguous()
        feats = self.pnet(keypoint_xyz, keypoint_features, grid)[1]            # (b, C, n*m)
        feats = feats.view(b, -1, n, m).permute(0, 2, 1, 3).contiguous().view(b, n, -1)
        return self.reduction(feats)
