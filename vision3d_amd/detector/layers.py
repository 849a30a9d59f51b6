"""Small dense building blocks of the detector.  Same classes, constructor arguments and state_dict keys as
vision3d/detector/layers.py:7-73 (VoxelFeatureExtractor, BEVFeatureGatherer, MLP)."""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


class VoxelFeatureExtractor(nn.Module):
    """(M, K, C) zero-padded point slots + (M,) occupancy -> (M, C) mean over the occupied slots.

    The device voxelizer emits this mean directly (`item['voxel_mean']`); the module exists for callers that hold
    the reference's `features` / `occupancy` pair."""

    def forward(self, feature, occupancy):
        count = occupancy.reshape(-1, 1).to(dtype=feature.dtype)
        return torch.div(feature.sum(dim=1), count).contiguous()


class BEVFeatureGatherer(nn.Module):
    """Bilinear lookup of BEV features at keypoint (x, y) positions.

    grid_sample wants coordinates in [-1, 1]; the reference's conversion (layers.py:29-44) clamps the fractional
    pixel index to [0, dim - 1] and then divides by (dim - 1) - 1, with W and H taken from the
    swapped axes of the spconv layout.  `_to_grid` reproduces exactly that arithmetic."""

    def __init__(self, cfg, voxel_offset, base_voxel_size):
        super().__init__()
        self.cfg = cfg
        # buffers of its own (the reference aliases the CNN's tensors, which then miss .cuda()/.to())
        self.register_buffer("pixel_offset", voxel_offset[:2].detach().clone(), persistent=False)
        self.register_buffer("base_pixel_size", base_voxel_size[:2].detach().clone(), persistent=False)

    def _limit(self, like, height, width):
        """[W - 1, H - 1] on the device, cached per map size: `new_tensor([...])` is a pageable host-to-device copy, i.e. a host
        synchronisation per frame (it serialised frames in flight on different streams)."""
        cache = self.__dict__.setdefault("_limit_cache", {})
        key = (int(height), int(width), str(like.device), like.dtype)
        if key not in cache:
            cache[key] = torch.tensor([width - 1, height - 1], dtype=like.dtype, device=like.device)
        return cache[key]

    def _to_grid(self, xy, height, width):
        pixel = self.base_pixel_size * self.cfg.STRIDES[-1]
        frac = (xy - self.pixel_offset) / pixel
        limit = self._limit(frac, height, width)
        frac = torch.min(frac.clamp(min=0), limit)
        return (2 * (frac / (limit - 1)) - 1).flip(-1)

    # reference method names, kept for callers that used them
    def normalize_indices(self, indices, H, W):
        limit = self._limit(indices, H, W)
        return 2 * (torch.min(indices.clamp(min=0), limit) / (limit - 1)) - 1

    def compute_bev_indices(self, keypoint_xyz, H, W):
        return self._to_grid(keypoint_xyz[:, None, :, :2], H, W)

    def forward(self, feature_map, keypoint_xyz):
        height, width = feature_map.shape[-2:]
        grid = self._to_grid(keypoint_xyz[:, None, :, :2], height, width)
        if (feature_map.is_cuda and feature_map.dtype == torch.float32 and feature_map.is_contiguous()
                and not (torch.is_grad_enabled() and (feature_map.requires_grad or grid.requires_grad))):
            from .. import _lib as L  # the same lookup, a thread per (keypoint, channel): csrc/pointops.hip
            b, c = feature_map.shape[:2]
            k = grid.shape[2]
            g = grid.reshape(b, k, 2).contiguous()
            out = torch.empty((b, c, k), dtype=torch.float32, device=feature_map.device)
            with torch.cuda.device(feature_map.device):
                L.check(L.lib().v3d_bev_bilinear(L.ptr(feature_map), L.ptr(g), b, c, height, width, k, L.ptr(out), L.stream_ptr()),
                        "bev_bilinear")
            return out
        return F.grid_sample(feature_map, grid, align_corners=True).squeeze(2)


def _per_layer(flag, n):
    return list(flag) if isinstance(flag, (list, tuple)) else [flag] * n


class MLP(nn.Sequential):
    """Stack of nn.Linear with optional bias / BatchNorm1d / ReLU per layer.  Children are named
    `linear_i`, `batchnorm_i`, `relu_i` -- the reference's state_dict keys (layers.py:53-73)."""

    def __init__(self, channels, bias=False, bn=False, relu=True):
        n = len(channels) - 1
        layers = OrderedDict()
        for i, (c_in, c_out, use_bias, use_bn, use_relu) in enumerate(
                zip(channels[:-1], channels[1:], _per_layer(bias, n), _per_layer(bn, n), _per_layer(relu, n))):
            linear = nn.Linear(c_in, c_out, bias=use_bias)
            nn.init.normal_(linear.weight, std=0.01)
            if use_bias:
                nn.init.zeros_(linear.bias)
            layers[f"linear_{i}"] = linear
            if use_bn:
                norm = nn.BatchNorm1d(c_out)
                nn.init.ones_(norm.weight)
                nn.init.zeros_(norm.bias)
                layers[f"batchnorm_{i}"] = norm
            if use_relu:
                layers[f"relu_{i}"] = nn.ReLU(inplace=True)
        super().__init__(layers)
