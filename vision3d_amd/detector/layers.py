"""Small dense layers of the detector (interface of vision3d/detector/layers.py:7-73)."""
import torch
from torch import nn
from torch.nn import functional as F


class VoxelFeatureExtractor(nn.Module):
    """Mean of the occupied point slots of each voxel: (N, K, C), (N,) -> (N, C) (layers.py:10-17).
    The device voxelizer already produces this mean (`item['voxel_mean']`); this module is the drop-in
    form for callers that hold `features`/`occupancy`."""

    def forward(self, feature, occupancy):
        return (feature.sum(1) / occupancy.to(feature.dtype).view(-1, 1)).contiguous()


class BEVFeatureGatherer(nn.Module):
    """Bilinear BEV feature lookup at keypoints (layers.py:20-50).  The index normalisation reproduces the
    reference verbatim, including its (dims - 1) divisor on already-decremented dims and the H/W swap
    (SURVEY.md H13)."""

    def __init__(self, cfg, voxel_offset, base_voxel_size):
        super().__init__()
        self.cfg = cfg
        # own buffers (not views of the CNN's): they must follow .cuda()/.to() with the module
        self.register_buffer("pixel_offset", voxel_offset[:2].detach().clone(), persistent=False)
        self.register_buffer("base_pixel_size", base_voxel_size[:2].detach().clone(), persistent=False)

    def normalize_indices(self, indices, H, W):
        dims = indices.new_tensor([W - 1, H - 1])
        clipped = torch.min(indices.clamp(min=0), dims)
        return 2 * (clipped / (dims - 1)) - 1

    def compute_bev_indices(self, keypoint_xyz, H, W):
        pix = (keypoint_xyz[:, None, :, :2] - self.pixel_offset) / (self.base_pixel_size * self.cfg.STRIDES[-1])
        return self.normalize_indices(pix, H, W).flip(3)

    def forward(self, feature_map, keypoint_xyz):
        _, _, H, W = feature_map.shape
        grid = self.compute_bev_indices(keypoint_xyz, H, W)
        return F.grid_sample(feature_map, grid, align_corners=True).squeeze(2)


class MLP(nn.Sequential):
    """Linear stack with optional per-layer bias / BatchNorm1d / ReLU; children are named
    linear_i / batchnorm_i / relu_i (layers.py:53-73) so state_dict keys match."""

    def __init__(self, channels, bias=False, bn=False, relu=True):
        super().__init__()
        n = len(channels) - 1
        bias, bn, relu = (v if isinstance(v, (list, tuple)) else [v] * n for v in (bias, bn, relu))
        for i in range(n):
            lin = nn.Linear(channels[i], channels[i + 1], bias=bias[i])
            nn.init.normal_(lin.weight, std=0.01)
            if bias[i]:
                nn.init.constant_(lin.bias, 0)
            self.add_module(f"linear_{i}", lin)
            if bn[i]:
                norm = nn.BatchNorm1d(channels[i + 1])
                nn.init.constant_(norm.weight, 1)
                nn.init.constant_(norm.bias, 0)
                self.add_module(f"batchnorm_{i}", norm)
            if relu[i]:
                self.add_module(f"relu_{i}", nn.ReLU(inplace=True))
