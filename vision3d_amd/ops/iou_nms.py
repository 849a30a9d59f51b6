"""Rotated BEV IoU / NMS -- the `vision3d.ops` names (vision3d/ops/iou_nms.py:9,38-134) on MI355X.

Boxes are (x_ctr, y_ctr, width, height, angle_degrees) float32.  The detector feeds yaw in radians to
these degree-based ops; that behaviour is part of the reference and is preserved.
All arithmetic runs in libvision3d_hip.so (csrc/iou_nms.hip); there is no CPU path.
"""
import numpy as np
import torch

from .. import _lib as L


def box_iou_rotated(boxes1, boxes2):
    """IoU matrix (M, N) float32.  Mirrors `vision3d._C.box_iou_rotated` (csrc/vision.cpp:63): float32
    only (box_iou_rotated_cuda.cu:68), inputs row-major contiguous."""
    L.require_gpu("box_iou_rotated", boxes1, boxes2)
    b1, b2 = L.as_f32("box_iou_rotated", boxes1), L.as_f32("box_iou_rotated", boxes2)
    if b1.dim() != 2 or b2.dim() != 2 or b1.shape[-1] != 5 or b2.shape[-1] != 5:
        raise RuntimeError("box_iou_rotated: expected (M,5) and (N,5)")
    m, n = b1.shape[0], b2.shape[0]
    out = torch.empty((m, n), dtype=torch.float32, device=b1.device)
    with L.device_guard(b1.device):
        L.check(L.lib().v3d_box_iou_rotated(L.ptr(b1), m, L.ptr(b2), n, L.ptr(out), L.stream_ptr()), "box_iou_rotated")
    return out


def box_iou_rotated_3d(boxes1, boxes2):
    """3-D IoU matrix (M, N) float32 of (x, y, z, w, l, h, yaw) boxes, z = centre: BEV intersection area -- the operator of
    `box_iou_rotated` on columns (0, 1, 3, 4, 6), so the yaw is read the way that operator reads it -- times the
    overlap of the z extents, over the union of the volumes.  Upstream declares the name and raises
    (vision3d/ops/iou_nms.py:12-13); this is the repository's definition (oracle/v3d_oracle.c:orc_box_iou_rotated_3d)."""
    L.require_gpu("box_iou_rotated_3d", boxes1, boxes2)
    b1, b2 = L.as_f32("box_iou_rotated_3d", boxes1), L.as_f32("box_iou_rotated_3d", boxes2)
    if b1.dim() != 2 or b2.dim() != 2 or b1.shape[-1] != 7 or b2.shape[-1] != 7:
        raise RuntimeError("box_iou_rotated_3d: expected (M,7) and (N,7)")
    m, n = b1.shape[0], b2.shape[0]
    out = torch.empty((m, n), dtype=torch.float32, device=b1.device)
    with L.device_guard(b1.device):
        L.check(L.lib().v3d_box_iou_rotated_3d(L.ptr(b1), m, L.ptr(b2), n, L.ptr(out), L.stream_ptr()), "box_iou_rotated_3d")
    return out


def _rule_threshold(iou_threshold, rule):
    """The kernels suppress on IoU >= t (the reference CPU path, nms_rotated_cpu.cpp:53 -- the parity target named by
    BASELINE.json: "match the reference CPU/PyTorch path").  The reference CUDA path suppresses on IoU > t
    (nms_rotated_cuda.cu:62-63); for float32 that is IoU >= nextafter(t, +inf), so rule="cuda" needs no second kernel.
    The two differ only when an IoU equals the threshold exactly."""
    t = np.float32(iou_threshold)
    if rule == "cpu":
        return float(t)
    if rule == "cuda":
        return float(np.nextafter(t, np.float32(np.inf)))
    raise ValueError("rule must be 'cpu' (IoU >= thr) or 'cuda' (IoU > thr)")


def nms_rotated_padded(boxes, scores, iou_threshold, rule="cpu"):
    """Device-resident result: (keep (N,) int64 padded, n_keep (1,) int32) -- no host synchronisation."""
    iou_threshold = _rule_threshold(iou_threshold, rule)
    L.require_gpu("nms_rotated", boxes, scores)
    b, s = L.as_f32("nms_rotated", boxes), L.as_f32("nms_rotated", scores)
    if b.dim() != 2 or b.shape[-1] != 5 or s.shape != (b.shape[0],):
        raise RuntimeError("nms_rotated: expected boxes (N,5) and scores (N,)")
    n = b.shape[0]
    keep = torch.empty((n,), dtype=torch.int64, device=b.device)
    n_keep = torch.zeros((1,), dtype=torch.int32, device=b.device)
    if n:
        lib = L.lib()
        ws = L.workspace(lib.v3d_nms_rotated_workspace(n), b.device)
        with L.device_guard(b.device):
            L.check(lib.v3d_nms_rotated(L.ptr(b), L.ptr(s), n, float(iou_threshold), L.ptr(keep), L.ptr(n_keep),
                                        L.ptr(ws), ws.numel(), L.stream_ptr()), "nms_rotated")
    return keep, n_keep


def nms_rotated(boxes, scores, iou_threshold, rule="cpu"):
    """Indices kept by greedy rotated NMS, by decreasing score (vision3d/ops/iou_nms.py:38-85).
    rule="cpu" (default): suppress on IoU >= threshold, the reference CPU path (nms_rotated_cpu.cpp:53) and this
    repository's parity target; rule="cuda": IoU > threshold, what the reference's CUDA build does (nms_rotated_cuda.cu:62)."""
    keep, n_keep = nms_rotated_padded(boxes, scores, iou_threshold, rule)
    return keep[: int(n_keep.item())]


def batched_nms_rotated(boxes, scores, idxs, iou_threshold, rule="cpu"):
    """Per-category NMS through the coordinate-offset trick (vision3d/ops/iou_nms.py:90-134): every
    category is shifted by idx * (max_coord - min_coord + 1) so categories never overlap."""
    assert boxes.shape[-1] == 5
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    hi = (torch.max(boxes[:, 0], boxes[:, 1]) + torch.max(boxes[:, 2], boxes[:, 3]) / 2).max()
    lo = (torch.min(boxes[:, 0], boxes[:, 1]) - torch.min(boxes[:, 2], boxes[:, 3]) / 2).min()
    shifted = boxes.clone()
    shifted[:, :2] += (idxs.to(boxes) * (hi - lo + 1))[:, None]
    return nms_rotated(shifted, scores, iou_threshold, rule)


def batched_nms_rotated_padded(boxes, scores, idxs, iou_threshold):
    """batched_nms_rotated without the host read: (keep (N,) int64 padded, n_keep (1,) int32), both on the
    device -- the form used inside captured HIP graphs."""
    assert boxes.shape[-1] == 5
    hi = (torch.max(boxes[:, 0], boxes[:, 1]) + torch.max(boxes[:, 2], boxes[:, 3]) / 2).max()
    lo = (torch.min(boxes[:, 0], boxes[:, 1]) - torch.min(boxes[:, 2], boxes[:, 3]) / 2).min()
    shifted = boxes.clone()
    shifted[:, :2] += (idxs.to(boxes) * (hi - lo + 1))[:, None]
    return nms_rotated_padded(shifted, scores, iou_threshold)


def nms(boxes, scores, iou_threshold):
    """Axis-aligned NMS on (x1,y1,x2,y2) (the torchvision name re-exported at iou_nms.py:6; never called
    by the detector).  Served by the rotated kernel with angle 0."""
    xy = (boxes[:, :2] + boxes[:, 2:4]) / 2
    wh = boxes[:, 2:4] - boxes[:, :2]
    rot = torch.cat((xy, wh, torch.zeros_like(xy[:, :1])), dim=1)
    return nms_rotated(rot, scores, iou_threshold)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Axis-aligned per-category NMS (iou_nms.py:16-33)."""
    assert boxes.shape[-1] == 4
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    offsets = idxs.to(boxes) * (boxes.max() + 1)
    return nms(boxes + offsets[:, None], scores, iou_threshold)
