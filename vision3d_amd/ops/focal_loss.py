"""Sigmoid focal loss (interface of `vision3d/ops/focal_loss.py:5-45`; RetinaNet, arXiv 1708.02002).

Elementwise and tiny next to the convolutions, so it stays in PyTorch.
"""
import torch
import torch.nn.functional as F


def sigmoid_focal_loss(inputs, targets, alpha: float = 0.25, gamma: float = 2, reduction: str = "none"):
    """FL = -alpha_t (1 - p_t)^gamma log(p_t) on logits `inputs`; alpha < 0 disables the class weight."""
    prob = torch.sigmoid(inputs)
    bce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    out = bce * (1 - p_t) ** gamma
    if alpha >= 0:
        out = (alpha * targets + (1 - alpha) * (1 - targets)) * out
    if reduction == "sum":
        return out.sum()
    if reduction == "mean":
        return out.mean()
    return out
