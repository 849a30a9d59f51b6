"""VoxelNet 7-DoF box parameterisation (interface of `vision3d/core/box_encode.py:13-36`).

decode: xyz = d_xyz * (diag, diag, h) + a_xyz ; wlh = exp(d_wlh) * a_wlh ; yaw = d_yaw + a_yaw
encode: inverse, with the yaw residual wrapped into [0, pi) (box_encode.py:34).
diag is the BEV diagonal sqrt(w^2 + l^2) of the anchor (box_encode.py:5-10).
"""
import math

import torch


def _scale(anchors):
    diag = torch.linalg.vector_norm(anchors[..., 3:5], dim=-1, keepdim=True)
    return torch.cat((diag, diag, anchors[..., 5:6]), dim=-1)


def decode(deltas, anchors):
    """deltas, anchors: (*, 7) -> boxes (*, 7)."""
    out = torch.empty_like(deltas)
    out[..., 0:3] = deltas[..., 0:3] * _scale(anchors) + anchors[..., 0:3]
    out[..., 3:6] = deltas[..., 3:6].exp() * anchors[..., 3:6]
    out[..., 6] = deltas[..., 6] + anchors[..., 6]
    return out


def encode(boxes, anchors):
    """boxes, anchors: (*, 7) -> deltas (*, 7)."""
    out = torch.empty_like(boxes)
    out[..., 0:3] = (boxes[..., 0:3] - anchors[..., 0:3]) / _scale(anchors)
    out[..., 3:6] = (boxes[..., 3:6] / anchors[..., 3:6]).log()
    out[..., 6] = torch.remainder(boxes[..., 6] - anchors[..., 6], math.pi)
    return out
