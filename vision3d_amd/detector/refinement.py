"""Refinement head (interface of vision3d/detector/refinement.py:8-50).  Upstream only the MLP forward
is meaningful: `apply_refinements` raises and `forward` splits with `split(1)` on dim 0, which works
only for batch size 2.  The evident intent -- 7 box deltas + 1 confidence on the last
dim -- is what `forward` returns here; `apply_refinements` is defined as the VoxelNet decoding the rest of the
reference uses for every box residual (core/box_encode.py:13-23), with the proposal in the anchor's role."""
import torch
from torch import nn

from ..core.box_encode import decode
from .layers import MLP


class RefinementLayer(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.mlp = self.build_mlp(cfg)

    def build_mlp(self, cfg):
        channels = cfg.REFINEMENT.MLPS + [cfg.BOX_DOF + 1]
        return MLP(channels, bias=True, bn=False, relu=[True, False])

    def apply_refinements(self, box_deltas, boxes):
        """(…, 7) residuals + (…, 7) proposals -> refined boxes: xyz = d * [diag, diag, h] + xyz_p, wlh = exp(d) * wlh_p,
        yaw = d + yaw_p.  Upstream raises (refinement.py:32-33); SURVEY.md 8(f) rank 3 asks for the decode of
        core/box_encode.py, which is what the encode side of refinement_targets.py would invert."""
        return decode(box_deltas, boxes)

    def forward(self, points, features, boxes):
        """features (B, N, C) pooled RoI features -> (box_deltas (B,N,7), scores (B,N,1))."""
        out = self.mlp(features)
        return out.split([self.cfg.BOX_DOF, 1], dim=-1)
