"""PV-RCNN pieces (interface of vision3d/detector/model.py:16-85): FPS keypoints, voxel-set
abstraction over the raw points + 4 sparse-CNN levels, BEV feature gather, proposal layer, RoI-grid
pooling, refinement MLP.  Upstream `forward` raises (stage 2 was never wired, model.py:84-85); the
stage-1 + feature pieces below are the ones that exist there, on the MI355X ops.
"""
from copy import deepcopy

import torch
from torch import nn

from ..pointnet2.pointnet2_modules import PointnetSAModuleMSG
from ..pointnet2.pointnet2_utils import furthest_point_sample, gather_operation
from .layers import BEVFeatureGatherer, VoxelFeatureExtractor
from .proposal import ProposalLayer
from .refinement import RefinementLayer
from .roi_grid_pool import RoiGridPool
from .sparse_cnn import CNN_FACTORY


class PV_RCNN(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.pnets = self.build_pointnets(cfg)
        self.roi_grid_pool = RoiGridPool(cfg)
        self.vfe = VoxelFeatureExtractor()
        self.cnn = CNN_FACTORY[cfg.CNN](cfg)
        self.bev = BEVFeatureGatherer(cfg, self.cnn.voxel_offset, self.cnn.base_voxel_size)
        self.proposal_layer = ProposalLayer(cfg)
        self.refinement_layer = RefinementLayer(cfg)

    def build_pointnets(self, cfg):
        nets = [PointnetSAModuleMSG(npoint=-1, radii=cfg.PSA.RADII[i], nsamples=cfg.SAMPLES_PN,
                                    mlps=deepcopy(mlps), use_xyz=True) for i, mlps in enumerate(cfg.PSA.MLPS)]
        return nn.Sequential(*nets)

    def sample_keypoints(self, points):
        """points (B, N, >=3) -> FPS keypoints (B, NUM_KEYPOINTS, 3)."""
        xyz = points[..., :3].contiguous()
        idx = furthest_point_sample(xyz, self.cfg.NUM_KEYPOINTS)
        return gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()

    def _pointnets(self, cnn_out, keypoint_xyz):
        outs = []
        for (xyz, feats), pnet in zip(cnn_out, self.pnets):
            outs.append(pnet(xyz.contiguous(), feats.transpose(1, 2).contiguous(), keypoint_xyz)[1])
        return outs

    def point_feature_extract(self, item, cnn_features, bev_map):
        raw = tuple(torch.split(item["points"], [3, 1], dim=-1))
        feats = self._pointnets([raw] + list(cnn_features), item["keypoints"])
        feats.append(self.bev(bev_map, item["keypoints"]))
        return torch.cat(feats, dim=1)

    def proposal(self, item):
        item["keypoints"] = self.sample_keypoints(item["points"])
        features = item["voxel_mean"] if "voxel_mean" in item else self.vfe(item["features"], item["occupancy"])
        cnn_features, bev_map = self.cnn(features, item["coordinates"], item["batch_size"])
        scores, boxes = self.proposal_layer(bev_map)
        item.update(dict(P_cls=scores, P_reg=boxes))
        item["_cnn_features"], item["_bev_map"] = cnn_features, bev_map
        return item

    def forward(self, item):
        raise NotImplementedError  # model.py:84-85
