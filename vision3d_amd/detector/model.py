"""PV-RCNN (interface of vision3d/detector/model.py:16-85).

Data flow of the pieces that exist upstream (its `forward` raises: stage 2 was never wired, model.py:84-85):

    points --FPS--> keypoints --------------------------------------------+
    voxels --sparse CNN--> 4 levels of (xyz, features) + BEV map          |
    [raw points, level 0..3] --set abstraction around the keypoints--> per-keypoint features, + bilinear BEV lookup
    BEV map --ProposalLayer--> (P_cls, P_reg)

Sub-module names (`pnets`, `roi_grid_pool`, `vfe`, `cnn`, `bev`, `proposal_layer`, `refinement_layer`) are the
reference's, so checkpoints load.  FPS / ball query / grouping run on the MI355X kernels behind
vision3d_amd.pointnet2.
"""
import copy

import torch
from torch import nn

from ..pointnet2 import pointnet2_utils as pn2
from ..pointnet2.pointnet2_modules import PointnetSAModuleMSG
from . import layers, proposal, refinement, roi_grid_pool, sparse_cnn


def _set_abstraction(radii, mlps, nsamples):
    # the SA module edits its channel lists in place (prepends the xyz channels): hand it a copy of the config's
    return PointnetSAModuleMSG(npoint=-1, radii=radii, nsamples=nsamples, mlps=copy.deepcopy(mlps), use_xyz=True)


class PV_RCNN(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.pnets = self.build_pointnets(cfg)
        self.roi_grid_pool = roi_grid_pool.RoiGridPool(cfg)
        self.vfe = layers.VoxelFeatureExtractor()
        self.cnn = sparse_cnn.CNN_FACTORY[cfg.CNN](cfg)
        self.bev = layers.BEVFeatureGatherer(cfg, self.cnn.voxel_offset, self.cnn.base_voxel_size)
        self.proposal_layer = proposal.ProposalLayer(cfg)
        self.refinement_layer = refinement.RefinementLayer(cfg)

    def build_pointnets(self, cfg):
        """One multi-scale set-abstraction module per feature source (raw points, then the CNN levels)."""
        return nn.Sequential(*(_set_abstraction(radii, mlps, cfg.SAMPLES_PN)
                               for radii, mlps in zip(cfg.PSA.RADII, cfg.PSA.MLPS)))

    def sample_keypoints(self, points):
        """points (B, N, >=3) -> (B, NUM_KEYPOINTS, 3): farthest-point sample of the xyz columns."""
        xyz = points[..., :3].contiguous()
        picked = pn2.furthest_point_sample(xyz, self.cfg.NUM_KEYPOINTS)           # (B, K) int32
        planes = pn2.gather_operation(xyz.transpose(1, 2).contiguous(), picked)   # (B, 3, K)
        return planes.transpose(1, 2).contiguous()

    def _pointnets(self, sources, keypoint_xyz):
        """sources: [(xyz (B, N, 3), features (B, N, C))] -> [(B, C_out, K)], one per source."""
        pooled = []
        for pnet, (xyz, features) in zip(self.pnets, sources):
            _, out = pnet(xyz.contiguous(), features.transpose(1, 2).contiguous(), keypoint_xyz)
            pooled.append(out)
        return pooled

    def point_feature_extract(self, item, cnn_features, bev_map):
        """Keypoint features: set abstraction over every source concatenated with the BEV lookup (B, C_total, K)."""
        keypoints = item["keypoints"]
        xyz, reflectance = item["points"].split([3, 1], dim=-1)
        pooled = self._pointnets([(xyz, reflectance), *cnn_features], keypoints)
        return torch.cat([*pooled, self.bev(bev_map, keypoints)], dim=1)

    def proposal(self, item):
        """Stage 1.  Adds keypoints, P_cls, P_reg (and the CNN outputs under `_cnn_features` / `_bev_map` for the
        keypoint-feature stage) to `item`."""
        item["keypoints"] = self.sample_keypoints(item["points"])
        if "voxel_mean" in item:      # device voxelizer output
            voxel_features = item["voxel_mean"]
        else:                         # reference-style (M, K, C) slots + occupancy
            voxel_features = self.vfe(item["features"], item["occupancy"])
        cnn_features, bev_map = self.cnn(voxel_features, item["coordinates"], item["batch_size"])
        item["P_cls"], item["P_reg"] = self.proposal_layer(bev_map)
        item["_cnn_features"], item["_bev_map"] = cnn_features, bev_map
        return item

    def forward(self, item):
        raise NotImplementedError("upstream never wired stage 2 into forward (vision3d/detector/model.py:84-85)")
