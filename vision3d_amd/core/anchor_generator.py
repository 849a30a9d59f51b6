"""Dense anchor grid, laid out (n_cls, n_yaw, ny, nx, 7) with rows (x, y, z, w, l, h, yaw).

Interface of `vision3d/core/anchor_generator.py:28-74` (`AnchorGenerator(cfg).anchors`); the tensor is
assembled directly in its final layout instead of cat+permute.  Cell centres sit at bin midpoints of
the BEV grid whose pixel is VOXEL_SIZE[:2] * STRIDES[-1] (anchor_generator.py:41-45).

Reference quirk reproduced on purpose: the reference writes the per-class centre z through an EXPANDED view
(`centers.expand(...)`, then `centers[:, :, :, arange(NUM_CLASSES), 2] = anchor_z`, anchor_generator.py:56-58) -- the class
dimension shares memory, so every class ends up with the LAST class's `center_z` (default 3-class configuration: -0.6 for
Car as well, not -1.0).  A reference-trained multi-class checkpoint was trained against that grid, so it is what this
generator returns; tests/golden/anchors.npz holds the reference's own output for 1 and 3 classes.
"""
import torch
from torch import nn


def bin_midpoints(lo, hi, n):
    """n samples at the centres of n equal bins on [lo, hi) -- float32 arithmetic on 0-dim tensors,
    matching anchor_generator.py:5-12 (the grid is compared with the reference's in tests/test_host_golden.py)."""
    lo = torch.as_tensor(lo, dtype=torch.float32)
    hi = torch.as_tensor(hi, dtype=torch.float32)
    width = (hi - lo) / n
    return torch.linspace(lo.item(), (hi - width).item(), int(n)) + width / 2


class AnchorGenerator(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.anchors = self.make_anchors()

    def compute_grid_params(self):
        pixel = torch.tensor(self.cfg.VOXEL_SIZE[:2]) * self.cfg.STRIDES[-1]
        bounds = torch.tensor(self.cfg.GRID_BOUNDS, dtype=torch.float32).view(2, 3)
        lower, upper = bounds[0, :2], bounds[1, :2]
        grid_shape = ((upper - lower) / pixel).long()
        return lower, upper, grid_shape

    def make_anchors(self):
        cfg = self.cfg
        lower, upper, (nx, ny) = self.compute_grid_params()
        nx, ny = int(nx), int(ny)
        xs = bin_midpoints(lower[0], upper[0], nx)
        ys = bin_midpoints(lower[1], upper[1], ny)
        n_cls = cfg.NUM_CLASSES
        n_yaw = len(cfg.ANCHORS[0]["yaw"])
        out = torch.empty(n_cls, n_yaw, ny, nx, 7, dtype=torch.float32)
        out[..., 0] = xs.view(1, 1, 1, nx)
        out[..., 1] = ys.view(1, 1, ny, 1)
        z_all = float(cfg.ANCHORS[n_cls - 1]["center_z"])  # the reference's expanded-view write: last class wins
        for c, spec in enumerate(cfg.ANCHORS[:n_cls]):
            out[c, ..., 2] = z_all
            out[c, ..., 3:6] = torch.tensor(spec["wlh"], dtype=torch.float32)
            out[c, ..., 6] = torch.tensor(spec["yaw"], dtype=torch.float32).view(n_yaw, 1, 1)
        return out.contiguous()
