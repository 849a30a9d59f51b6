"""GT-sampling + global augmentation with the scene resident on the GPU (interface of vision3d/dataset/augmentation.py:18-198).

    aug = ChainedAugmentation(cfg)                       # loads cfg.DATA.CACHEDIR/database.pkl like the reference, or
    aug = ChainedAugmentation(cfg, database=db)          # a {class: [dict(points, box), ...]} dict / a SampleDatabase
    points, boxes, class_idx = aug(points, boxes, class_idx)

What runs where.  The reference does all of this in numpy inside DataLoader workers and goes to the GPU only for the
collision IoU.  Here the cloud, the sample database and the boxes stay on the device: the collision filter is the
(n + k)^2 rotated-IoU kernel, "points not in the pasted rectangles" is the points-in-boxes kernel, gathering and
translating the sampled objects' points are index operations on one concatenated database tensor.  One host read per
frame (the collision mask: the result is ragged).  The random draws are scalars and stay on the host, made through the
SAME numpy calls in the SAME order as the reference (`rng` defaults to the global `numpy.random`), so a seeded run
reproduces the reference draw for draw -- the golden test relies on that.

Arithmetic follows numpy's promotions, because they decide the values: pasted boxes and points are float32 + float64
positions (augmentation.py:162-166), so from there on the reference computes in float64 and casts to float32 at the very
end (kitti_dataset.py:119-120); without sampling everything stays float32.  The same dtypes are used on the device.

Two forms of the same computation.  The classes `SampleAugmentation` / `FlipAugmentation` / `ScaleAugmentation` /
`RotateAugmentation` mirror the reference's classes one to one on torch device tensors (about 60 small launches per frame).
`ChainedAugmentation` runs the whole chain through `v3d_augment_frame` (csrc/augment.hip): none of the draws depends on a device
result, so all of them are made first -- same calls, same order -- and the frame is two launches (collision filter + boxes; point
filter + ordered compaction + paste + transform) and one host read (the ragged sizes).  `ChainedAugmentation(fused=False)` keeps
the class-by-class chain; the two are compared bit for bit in tests/test_gpu_augmentation.py.
"""
import os
import pickle

import numpy as np
import torch

from .. import _lib as L
from ..core.geometry import PointsNotInRectangles
from ..ops import box_iou_rotated

BEV = [0, 1, 3, 4, 6]


def _to_device(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
    t = t.cuda() if not t.is_cuda else t
    return t if dtype is None else t.to(dtype)


class SampleDatabase:
    """The sample database on the device: per class ONE concatenated (sum n_i, 4) float32 point tensor, the segment offsets
    and the (K, 7) boxes (xy demeaned: box centre at the origin, augmentation.py:214-222)."""

    def __init__(self, database, num_classes):
        self.points, self.offsets, self.sizes, self.boxes = [], [], [], []
        for c in range(num_classes):
            items = database.get(c, []) if isinstance(database, dict) else database[c]
            sizes = np.array([len(it["points"]) for it in items], np.int64)
            pts = np.concatenate([np.asarray(it["points"], np.float32) for it in items]) if len(items) else np.zeros((0, 4), np.float32)
            box = np.stack([np.asarray(it["box"], np.float32) for it in items]) if len(items) else np.zeros((0, 7), np.float32)
            self.sizes.append(sizes)
            self.offsets.append(np.concatenate([[0], np.cumsum(sizes)]))
            self.points.append(torch.from_numpy(pts).cuda())
            self.boxes.append(torch.from_numpy(box).cuda())

    def __len__(self):
        return len(self.points)

    def count(self, class_idx):
        return len(self.sizes[class_idx])

    def flat(self):
        """-> (all points (sum P, 4), all boxes (sum K, 7), first point row per class, first box row per class): the per-class
        tensors behind each other, what `v3d_augment_frame` indexes.  Built on first use."""
        if getattr(self, "_flat", None) is None:
            dev = self.points[0].device if self.points else torch.device("cuda")
            pts = torch.cat(self.points) if self.points else torch.zeros((0, 4), device=dev)
            box = torch.cat(self.boxes) if self.boxes else torch.zeros((0, 7), device=dev)
            pt_base = np.concatenate([[0], np.cumsum([p.shape[0] for p in self.points])]).astype(np.int64)
            box_base = np.concatenate([[0], np.cumsum([b.shape[0] for b in self.boxes])]).astype(np.int64)
            self._flat = (pts.contiguous(), box.contiguous(), pt_base, box_base)
        return self._flat


class Augmentation:

    def __init__(self, cfg, rng=None):
        self.cfg = cfg
        self.rng = rng if rng is not None else np.random  # anything with choice / rand / uniform

    def uniform(self, *args):
        return np.float32(self.rng.uniform(*args))

    def __call__(self, points, boxes, *args):
        raise NotImplementedError


class RotateAugmentation(Augmentation):
    """Rotation about z by theta ~ U(GLOBAL_ROTATION) (augmentation.py:51-75): xy <- xy @ [[c, s], [-s, c]], yaw += theta."""

    @staticmethod
    def rotate(theta, xy):
        c, s = float(np.cos(theta)), float(np.sin(theta))  # float32 cos/sin of the float32 angle, as the reference's matrix
        x, y = xy[:, 0], xy[:, 1]
        return torch.stack((x * c + y * (-s), x * s + y * c), 1)

    def __call__(self, points, boxes):
        theta = self.uniform(*self.cfg.AUG.GLOBAL_ROTATION)
        points = torch.cat((self.rotate(theta, points[:, :2]), points[:, 2:]), 1)
        boxes = torch.cat((self.rotate(theta, boxes[:, :2]), boxes[:, 2:6], boxes[:, 6:] + float(theta)), 1)
        return points, boxes


class FlipAugmentation(Augmentation):
    """y <- -y, yaw <- -yaw with probability 1/2 (augmentation.py:78-95)."""

    def __call__(self, points, boxes):
        if self.rng.rand() < 0.5 or not self.cfg.AUG.FLIP_HORIZONTAL:
            return points, boxes
        sign_p = points.new_tensor([1, -1] + [1] * (points.shape[1] - 2))
        sign_b = boxes.new_tensor([1, -1, 1, 1, 1, 1, -1])
        return points * sign_p, boxes * sign_b


class ScaleAugmentation(Augmentation):
    """xyz and (x, y, z, w, l, h) times a factor ~ U(GLOBAL_SCALE) (augmentation.py:98-114)."""

    def __call__(self, points, boxes):
        factor = float(self.uniform(*self.cfg.AUG.GLOBAL_SCALE))
        points = torch.cat((points[:, :3] * factor, points[:, 3:]), 1)
        boxes = torch.cat((boxes[:, :6] * factor, boxes[:, 6:]), 1)
        return points, boxes


class SampleAugmentation(Augmentation):
    """Pastes database objects into the scene (augmentation.py:117-198): draw NUM_SAMPLE_OBJECTS per class, move each to a
    uniform position inside the grid bounds, drop the ones whose BEV rectangle overlaps anything else (IoU > 1e-2 with any
    other box, scene or sample), remove the scene points under the pasted rectangles, append the samples."""

    def __init__(self, cfg, database=None, rng=None):
        super().__init__(cfg, rng)
        if database is None:
            with open(os.path.join(cfg.DATA.CACHEDIR, "database.pkl"), "rb") as f:
                database = pickle.load(f)
        self.database = database if isinstance(database, SampleDatabase) else SampleDatabase(database, cfg.NUM_CLASSES)

    def draw_samples(self):
        """[(class, index)] -- the reference's draws, class by class."""
        picks = []
        for c in range(self.cfg.NUM_CLASSES):
            idx = self.rng.choice(self.database.count(c), self.cfg.AUG.NUM_SAMPLE_OBJECTS[c]).tolist()
            picks += [(c, i) for i in idx]
        return picks

    def gather(self, picks):
        """-> sample boxes (k, 7) float32, their points concatenated (P, 4) float32, segment id per point (P,), class per sample."""
        db = self.database
        boxes, points, seg, cls, k = [], [], [], [], 0
        for c in sorted({c for c, _ in picks}):
            idx = np.array([i for cc, i in picks if cc == c], np.int64)
            lengths = db.sizes[c][idx]
            starts = db.offsets[c][idx]
            within = np.arange(lengths.sum()) - np.repeat(np.cumsum(lengths) - lengths, lengths)
            src = torch.from_numpy(np.repeat(starts, lengths) + within).cuda()
            points.append(db.points[c][src])
            seg.append(torch.from_numpy(np.repeat(np.arange(k, k + len(idx)), lengths)).cuda())
            boxes.append(db.boxes[c][torch.from_numpy(idx).cuda()])
            cls += [c] * len(idx)
            k += len(idx)
        return torch.cat(boxes), torch.cat(points), torch.cat(seg), np.array(cls, np.int64)

    def filter_collisions(self, boxes, sample_boxes):
        """(k,) bool on the device: samples whose only BEV overlap (IoU > 1e-2) is with themselves (augmentation.py:140-149)."""
        n = boxes.shape[0]
        bev = torch.cat((boxes.double(), sample_boxes.double())).float()[:, BEV].contiguous()
        iou = box_iou_rotated(bev, bev)
        return (iou > 1e-2).sum(1)[n:] == 1

    def __call__(self, points, boxes, class_idx):
        picks = self.draw_samples()
        if not picks:  # (the reference's np.stack of an empty list raises here)
            return points, boxes, class_idx
        sample_boxes, sample_points, seg, sample_cls = self.gather(picks)
        lower, upper = np.r_[self.cfg.GRID_BOUNDS].reshape(2, 3)[:, :2]
        position = torch.from_numpy(self.rng.rand(len(picks), 2) * (upper - lower) + lower).cuda()  # float64
        sample_boxes = sample_boxes.double()
        sample_boxes[:, :2] += position
        sample_points = sample_points.double()
        sample_points[:, :2] += position[seg]
        keep = self.filter_collisions(boxes, sample_boxes)
        keep_host = keep.cpu().numpy()  # the one host read: what follows is ragged
        sample_boxes = sample_boxes[keep]
        sample_points = sample_points[keep[seg]]
        points = PointsNotInRectangles(points)(sample_boxes.float())
        points = torch.cat((points.double(), sample_points))
        boxes = torch.cat((boxes.double(), sample_boxes))
        class_idx = torch.cat((class_idx, torch.from_numpy(sample_cls[keep_host]).to(class_idx.device)))
        return points, boxes, class_idx


SAMPLE_RECORD = np.dtype([("box_row", np.int32), ("pt_start", np.int32), ("pt_len", np.int32), ("cls", np.int32),
                          ("px", np.float64), ("py", np.float64)])  # AugSample of csrc/augment.hip


class ChainedAugmentation(Augmentation):
    """sample -> flip -> scale -> rotate (augmentation.py:31-48).  numpy in -> numpy out (float32, the dtype the reference's
    dataset casts to); cuda tensors in -> cuda tensors out.  fused (default): the frame through `v3d_augment_frame`."""

    def __init__(self, cfg, database=None, rng=None, fused=True):
        super().__init__(cfg, rng)
        self.fused = fused
        self.sample = SampleAugmentation(cfg, database, self.rng) if cfg.AUG.DATABASE_SAMPLE else None
        self.augmentations = [FlipAugmentation(cfg, self.rng), ScaleAugmentation(cfg, self.rng), RotateAugmentation(cfg, self.rng)]

    def __call__(self, points, boxes, class_idx):
        as_numpy = isinstance(points, np.ndarray)
        points, boxes = _to_device(points, torch.float32), _to_device(boxes, torch.float32)
        class_idx = _to_device(np.asarray(class_idx, np.int64) if as_numpy else class_idx, torch.int64)
        if self.fused:
            points, boxes, class_idx = self.fused_frame(points, boxes, class_idx)
        else:
            if self.sample is not None:
                points, boxes, class_idx = self.sample(points, boxes, class_idx)
            for aug in self.augmentations:
                points, boxes = aug(points, boxes)
            points, boxes = points.float(), boxes.float()
        if as_numpy:
            return points.cpu().numpy(), boxes.cpu().numpy(), class_idx.cpu().numpy()
        return points, boxes, class_idx

    def draw(self):
        """Every random draw of one frame, through the calls the chained classes make, in their order: the samples class by
        class (augmentation.py:132), their positions (:164), the flip coin (:90), the scale (:110), the angle (:71)."""
        cfg, picks, position = self.cfg, [], None
        if self.sample is not None:
            picks = self.sample.draw_samples()
            if picks:
                lower, upper = np.r_[cfg.GRID_BOUNDS].reshape(2, 3)[:, :2]
                position = self.rng.rand(len(picks), 2) * (upper - lower) + lower  # float64
        flip = not (self.rng.rand() < 0.5 or not cfg.AUG.FLIP_HORIZONTAL)
        factor = float(self.uniform(*cfg.AUG.GLOBAL_SCALE))
        theta = self.uniform(*cfg.AUG.GLOBAL_ROTATION)
        return picks, position, flip, factor, theta

    def sample_records(self, picks, position):
        db = self.sample.database
        _, _, pt_base, box_base = db.flat()
        rec = np.zeros(len(picks), SAMPLE_RECORD)
        for j, (c, i) in enumerate(picks):
            rec[j] = (box_base[c] + i, pt_base[c] + db.offsets[c][i], db.sizes[c][i], c, position[j, 0], position[j, 1])
        return rec

    def fused_frame(self, points, boxes, class_idx):
        L.require_gpu("augment_frame", points, boxes, class_idx)
        picks, position, flip, factor, theta = self.draw()
        points, boxes, class_idx = points.contiguous(), boxes.contiguous(), class_idx.contiguous()
        N, C = points.shape
        n, k = boxes.shape[0], len(picks)
        cos_t, sin_t = float(np.cos(theta)), float(np.sin(theta))  # float32 cos / sin of the float32 angle, as the reference's matrix
        dev = points.device
        with torch.cuda.device(dev):
            if k == 0:
                out_p, out_b = torch.empty_like(points), torch.empty_like(boxes)
                L.check(L.lib().v3d_augment_frame(L.ptr(points), N, C, L.ptr(boxes), 0, n, 0, 0, 0, 0, 0, int(flip), factor, cos_t, sin_t,
                                                  float(theta), L.ptr(out_p), L.ptr(out_b), 0, 0, 0, L.stream_ptr()), "augment_frame")
                return out_p, out_b, class_idx
            if C != 4:
                raise RuntimeError(f"augment_frame: the sample database holds 4-column points, the scene has {C}")
            db_points, db_boxes, _, _ = self.sample.database.flat()
            rec = self.sample_records(picks, position)
            sample_points = int(rec["pt_len"].sum())
            samples = torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(dev)
            work_bytes = int(L.lib().v3d_augment_work_bytes(N, n, k))
            work = torch.empty((work_bytes + 7) // 8, dtype=torch.int64, device=dev)
            out_p = torch.empty((N + sample_points, 4), dtype=torch.float32, device=dev)
            out_b = torch.empty((n + k, 7), dtype=torch.float32, device=dev)
            out_c = torch.empty(n + k, dtype=torch.int64, device=dev)
            L.check(L.lib().v3d_augment_frame(L.ptr(points), N, C, L.ptr(boxes), L.ptr(class_idx), n, L.ptr(db_points), L.ptr(db_boxes),
                                              L.ptr(samples), k, sample_points, int(flip), factor, cos_t, sin_t, float(theta),
                                              L.ptr(out_p), L.ptr(out_b), L.ptr(out_c), L.ptr(work), work_bytes, L.stream_ptr()),
                    "augment_frame")
            head = work[:2].view(torch.int32).cpu().numpy()  # the one host read: {kept samples, their points, kept scene points}
        kept, kept_points, scene_points = int(head[0]), int(head[1]), int(head[2])
        return out_p[:scene_points + kept_points], out_b[:n + kept], out_c[:n + kept]
