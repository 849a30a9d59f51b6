"""RoI-grid pooling: random points inside every proposal, multi-scale set abstraction of the keypoint features around
them (ball query + group on the MI355X kernels), flatten per proposal, reduce with an MLP.  Interface of
vision3d/detector/roi_grid_pool.py:10-72 (`RoiGridPool(cfg)(proposals, keypoint_xyz, keypoint_features)`).

The reference draws its points with an unseeded torch.rand (roi_grid_pool.py:59); here the draw can be
injected (`samples`) or seeded (`self.generator`) so that results are reproducible.
"""
import copy

import torch
from torch import nn

from ..pointnet2.pointnet2_modules import PointnetSAModuleMSG
from .layers import MLP


def yaw_rotate(local, yaw):
    """local (..., m, 3) offsets in the box frame, yaw (...) radians -> offsets in the world frame (rotation about z)."""
    cos, sin = yaw.cos().unsqueeze(-1), yaw.sin().unsqueeze(-1)
    lx, ly, lz = local[..., 0], local[..., 1], local[..., 2]
    return torch.stack((cos * lx - sin * ly, sin * lx + cos * ly, lz), dim=-1)


class RoiGridPool(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        grid = cfg.GRIDPOOL
        # the SA module appends 3 to the first channel count in place: give it a private copy of the spec
        self.pnet = PointnetSAModuleMSG(npoint=-1, radii=grid.RADII_PN, nsamples=cfg.SAMPLES_PN,
                                        mlps=copy.deepcopy(grid.MLPS_PN), use_xyz=True)
        self.reduction = MLP(grid.MLPS_REDUCTION)
        self.generator = None

    def build_pointnet(self, cfg):  # reference entry point; the constructor builds the module inline
        return PointnetSAModuleMSG(npoint=-1, radii=cfg.GRIDPOOL.RADII_PN, nsamples=cfg.SAMPLES_PN,
                                   mlps=copy.deepcopy(cfg.GRIDPOOL.MLPS_PN), use_xyz=True)

    rotate_z = staticmethod(yaw_rotate)
    TORCH_TRIG = False  # sample_gridpoints: cos / sin of the yaw from torch (two more launches) instead of inside v3d_roi_grid_points

    def sample_gridpoints(self, boxes, samples=None):
        """boxes (b, n, 7) -> (b, n, m, 3) points uniform in each box: unit-cube draws scaled by (w, l, h), rotated
        by the yaw, moved to the centre."""
        b, n = boxes.shape[:2]
        if samples is None:
            samples = torch.rand((b, n, self.cfg.GRIDPOOL.NUM_GRIDPOINTS, 3), device=boxes.device, generator=self.generator)
        if (boxes.is_cuda and boxes.dtype == torch.float32 and samples.dtype == torch.float32 and boxes.shape[-1] == 7
                and not (torch.is_grad_enabled() and (boxes.requires_grad or samples.requires_grad))):
            # the statements below in one launch (csrc/pointops.hip v3d_roi_grid_points): same values.  The yaw's cos / sin are the
            # device library's cosf / sinf inside that launch -- on this stack bit-identical to torch.cos / torch.sin
            # (tests/test_gpu_pointops.py::test_roi_grid_points_trig_equals_torch); TORCH_TRIG: hand torch's values in instead
            from .. import _lib as L
            bx, sm = boxes.contiguous(), samples.contiguous()
            cos = sin = None
            if self.TORCH_TRIG:
                yaw = bx[..., 6]
                cos, sin = yaw.cos().contiguous(), yaw.sin().contiguous()
            out = torch.empty_like(sm)
            with L.device_guard(bx.device):
                L.check(L.lib().v3d_roi_grid_points(L.ptr(bx), L.ptr(sm), L.ptr(cos), L.ptr(sin), b * n, sm.shape[2], L.ptr(out),
                                                    L.stream_ptr()), "roi_grid_points")
            return out
        centre, size, yaw = boxes[..., None, 0:3], boxes[..., None, 3:6], boxes[..., 6]
        return centre + yaw_rotate(size * (samples - 0.5), yaw)

    def sample_gridpoints_torch(self, boxes, samples):
        """The reference's statements op by op (the cross-check of the fused launch in the tests)."""
        centre, size, yaw = boxes[..., None, 0:3], boxes[..., None, 3:6], boxes[..., 6]
        return centre + yaw_rotate(size * (samples - 0.5), yaw)

    def forward(self, proposals, keypoint_xyz, keypoint_features, samples=None):
        b, n = proposals.shape[:2]
        points = self.sample_gridpoints(proposals, samples)
        m = points.shape[2]
        new_xyz = points.reshape(b, n * m, 3).contiguous()
        pm = keypoint_features.transpose(1, 2)  # (b, K, C): contiguous when the features came from PV_RCNN's fused extraction
        if self.pnet._fusable(pm) and self.reduction.native_ok(pm):
            # inference: the pooled rows stay point-major -- (b, n * m, C) read as (b, n, m * C) IS the per-box row, in (grid point,
            # channel) order instead of the reference's (channel, grid point): the first reduction layer's weight rows are permuted
            # once instead of the activations every frame (roi_grid_pool.py:64-72)
            grid = getattr(keypoint_features, "_v3d_keypoint_grid", None)  # built with the frame's other grids (PV_RCNN._point_features_fused)
            kp = keypoint_xyz.contiguous()
            if grid is not None and not grid.matches(kp, self.pnet.max_radius()):
                grid = None
            pooled = self.pnet.fused_forward(kp, pm, new_xyz, grid=grid)  # (b, n * m, C)
            c = pooled.shape[2]
            return self.reduction.native_forward(pooled.reshape(b, n, m * c), first_rows=self._first_rows(m, c, pooled.device))
        _, pooled = self.pnet(keypoint_xyz, keypoint_features, new_xyz)  # (b, C, n*m)
        per_box = pooled.reshape(b, -1, n, m).permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.reduction(per_box)

    def _first_rows(self, m, c, device):
        """Row j * C + ch of the point-major per-box vector is feature ch * m + j of the reference's (channel, grid point) order."""
        cache = self.__dict__.setdefault("_rows_cache", {})
        key = (m, c, str(device))
        if key not in cache:
            j, ch = torch.meshgrid(torch.arange(m), torch.arange(c), indexing="ij")
            cache[key] = (ch * m + j).reshape(-1).to(device)
        return cache[key]
