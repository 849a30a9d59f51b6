"""Batch preprocessing with the voxelizer ON THE DEVICE (interface of vision3d/core/preprocess.py:10-79).

`Preprocessor(cfg)(item)` consumes `item['points']` = list of per-frame (Np, C) clouds and adds
  points (B, Np_max, C) f32 | features (M, K, C) f32 | coordinates (M, 4) i32 (b,z,y,x) |
  occupancy (M,) i32 | batch_size int            (all tensors already on the GPU)
-- the same keys/shapes the reference produces on the host (preprocess.py:47-61); the detector's
`.cuda()` calls on them become no-ops.  Voxel order/content equals the sequential reference loop
(first-touch order, first-come slots).  Extra key `voxel_mean` (M, C) carries the fused
VoxelFeatureExtractor output for callers that want to skip `features`.
"""
from collections import defaultdict

import numpy as np
import torch
from torch import nn

from ..spconv.utils import VoxelGenerator, voxelize_batch


class Preprocessor(nn.Module):

    def __init__(self, cfg, seed=None):
        super().__init__()
        self.cfg = cfg
        self.voxel_generator = self.build_voxel_generator(cfg)
        self._rng = np.random.default_rng(seed)  # reference pads with unseeded np.random (H12)

    def build_voxel_generator(self, cfg):
        return VoxelGenerator(voxel_size=cfg.VOXEL_SIZE, point_cloud_range=cfg.GRID_BOUNDS,
                              max_voxels=cfg.MAX_VOXELS, max_num_points=cfg.MAX_OCCUPANCY)

    @staticmethod
    def _to_device(p):
        if isinstance(p, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)).cuda(non_blocking=True)
        return p.cuda()

    def generate_batch_voxels(self, points):
        """One fused launch sequence for the whole batch (the reference loops frames on the host)."""
        offsets = np.concatenate([[0], np.cumsum([p.shape[0] for p in points])]).astype(np.int64).tolist()
        flat = torch.cat(points, dim=0) if len(points) > 1 else points[0]
        voxels, coords, occ, mean, n_vox = voxelize_batch(flat, offsets, self.cfg.VOXEL_SIZE, self.cfg.GRID_BOUNDS,
                                                          self.cfg.MAX_OCCUPANCY, self.cfg.MAX_VOXELS)
        m = int(n_vox.item())
        return voxels[:m], coords[:m], occ[:m], mean[:m]

    def pad_for_batch(self, points):
        """Dense (B, N, C) minibatch; short frames are padded by resampling their own points."""
        n_max = max(p.shape[0] for p in points)
        rows = []
        for p in points:
            pad = n_max - p.shape[0]
            if pad:
                idx = torch.from_numpy(self._rng.integers(0, p.shape[0], pad)).to(p.device)
                p = torch.cat((p, p[idx]))
            rows.append(p)
        return torch.stack(rows, dim=0)

    def forward(self, item):
        points = [self._to_device(p) for p in item["points"]]
        features, coordinates, occupancy, mean = self.generate_batch_voxels(points)
        item.update(points=self.pad_for_batch(points), features=features, coordinates=coordinates,
                    occupancy=occupancy, voxel_mean=mean, batch_size=len(points))
        return item


class TrainPreprocessor(Preprocessor):

    def collate_mapping(self, key, val):
        if key in ("G_cls", "G_reg", "M_cls", "M_reg"):
            return torch.stack(val)
        return val

    def collate(self, items):
        batch = defaultdict(list)
        for it in items:
            for key, val in it.items():
                batch[key].append(val)
        return self({k: self.collate_mapping(k, v) for k, v in batch.items()})
