"""Points-in-boxes on the device (interface of vision3d/core/geometry.py:27-65).

`PointsInCuboids(points)(boxes)` -> list of per-box point arrays; `PointsNotInRectangles(points)(boxes)`
-> points outside every BEV rectangle.  points (N, >=3), boxes (n, 7) = (x,y,z,w,l,h,yaw).  numpy inputs
are accepted (moved to the GPU) and numpy is returned for them, matching the reference call sites
(dataset/augmentation.py:195,233).
"""
import numpy as np
import torch

from .. import _lib as L


def points_in_boxes_mask(points, boxes, use_z=True):
    """(N, n) bool mask on the device (csrc/iou_nms.hip points_in_boxes_kernel)."""
    L.require_gpu("points_in_boxes", points, boxes)
    p = L.as_f32("points_in_boxes", points)
    b = L.as_f32("points_in_boxes", boxes)
    n, c = p.shape
    nb = b.shape[0]
    mask = torch.empty((n, nb), dtype=torch.uint8, device=p.device)
    with torch.cuda.device(p.device):
        L.check(L.lib().v3d_points_in_boxes(L.ptr(p), n, c, L.ptr(b), nb, int(bool(use_z)), L.ptr(mask), L.stream_ptr()),
                "points_in_boxes")
    return mask.bool()


def _to_dev(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda(), True
    return x, False


class PointsInCuboids:

    use_z = True

    def __init__(self, points):
        self.points, self._numpy = _to_dev(points)

    def _get_mask(self, boxes):
        boxes, _ = _to_dev(boxes)
        return points_in_boxes_mask(self.points, boxes, self.use_z)

    def __call__(self, boxes):
        mask = self._get_mask(boxes).T
        out = [self.points[m] for m in mask]
        return [o.cpu().numpy() for o in out] if self._numpy else out


class PointsNotInRectangles(PointsInCuboids):

    use_z = False

    def __call__(self, boxes):
        keep = ~self._get_mask(boxes).any(1)
        out = self.points[keep]
        return out.cpu().numpy() if self._numpy else out
