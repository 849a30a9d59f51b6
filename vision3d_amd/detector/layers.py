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

    def _native(self, feature_map, keypoint_xyz):
        return (feature_map.is_cuda and feature_map.dtype == torch.float32 and feature_map.is_contiguous()
                and keypoint_xyz.dtype == torch.float32
                and not (torch.is_grad_enabled() and (feature_map.requires_grad or keypoint_xyz.requires_grad)))

    def gather_point_major(self, feature_map, keypoint_xyz, out_pm=None):
        """The whole forward in ONE launch (csrc/pointops.hip v3d_bev_gather_keypoints: the statements of `_to_grid` on the device, one
        IEEE operation each, then the bilinear lookup): -> (B, K, C) POINT-major, or written into `out_pm` (a (B, K, >= C) view with
        unit channel stride, frames back to back: a column block of the keypoint feature matrix).  Same values as `forward`."""
        from .. import _lib as L
        b, c, height, width = feature_map.shape
        xyz = keypoint_xyz.contiguous()
        k = xyz.shape[1]
        consts = self.__dict__.setdefault("_host_consts", {})
        key = (str(feature_map.device),)
        if key not in consts:  # the module's own fp32 arithmetic, once, on the host: pixel = base_pixel_size * STRIDES[-1]
            pixel = (self.base_pixel_size.detach().cpu().float() * self.cfg.STRIDES[-1]).tolist()
            consts[key] = (self.pixel_offset.detach().cpu().float().tolist(), pixel)
        (off_x, off_y), (pix_x, pix_y) = consts[key]
        if out_pm is None:
            out_pm = torch.empty((b, k, c), dtype=torch.float32, device=feature_map.device)
        if out_pm.shape[:2] != (b, k) or out_pm.shape[2] < c or out_pm.stride(2) != 1 or out_pm.stride(0) != k * out_pm.stride(1):
            raise RuntimeError("gather_point_major: out_pm must be a (B, K, >= C) view with unit channel stride and frames back to back")
        with L.device_guard(feature_map.device):
            L.check(L.lib().v3d_bev_gather_keypoints(L.ptr(feature_map), L.ptr(xyz), b, c, height, width, k, off_x, off_y, pix_x, pix_y,
                                                     L.ptr(out_pm), out_pm.stride(1), L.stream_ptr()), "bev_gather_keypoints")
        return out_pm[:, :, :c]

    def forward(self, feature_map, keypoint_xyz):
        if self._native(feature_map, keypoint_xyz):
            return self.gather_point_major(feature_map, keypoint_xyz).transpose(1, 2)  # (B, C, K), a view of point-major rows
        height, width = feature_map.shape[-2:]
        grid = self._to_grid(keypoint_xyz[:, None, :, :2], height, width)
        return F.grid_sample(feature_map, grid, align_corners=True).squeeze(2)

    def forward_torch(self, feature_map, keypoint_xyz):
        """The reference's statements op by op (torch elementwise launches + v3d_bev_bilinear / grid_sample): the cross-check of the
        fused launch in the tests."""
        height, width = feature_map.shape[-2:]
        grid = self._to_grid(keypoint_xyz[:, None, :, :2], height, width)
        if self._native(feature_map, keypoint_xyz):
            from .. import _lib as L  # the same lookup, a thread per (keypoint, channel): csrc/pointops.hip
            b, c = feature_map.shape[:2]
            k = grid.shape[2]
            g = grid.reshape(b, k, 2).contiguous()
            out = torch.empty((b, c, k), dtype=torch.float32, device=feature_map.device)
            with L.device_guard(feature_map.device):
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

    # ---- inference on csrc/sa_mlp.hip linear_rows_kernel: the reduction / refinement MLPs have a hundred rows (one per proposal) and
    #      ran as four library GEMM launches + bias / ReLU launches; one launch per layer, exact fp32 products, fixed summation order
    def native_ok(self, x):
        if self.training or torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32:
            return False
        mods = list(self)
        if any(not isinstance(m, (nn.Linear, nn.ReLU)) for m in mods):  # (BatchNorm1d layers: the torch modules)
            return False
        return all(m.in_features % 4 == 0 for m in mods if isinstance(m, nn.Linear))

    def _packed(self, first_rows=None):
        """[(W^T (K, Nout padded to 16), bias padded or None, relu, Nout)] per Linear, cached until a parameter changes.  `first_rows`:
        a permutation of the FIRST layer's input features (the caller holds them in another order: RoiGridPool's point-major rows)."""
        mods = list(self)
        lins = [m for m in mods if isinstance(m, nn.Linear)]
        stamp = tuple((t.data_ptr(), t._version) for l in lins for t in (l.weight, l.bias) if t is not None)
        cache = self.__dict__.setdefault("_pack_cache", {})
        key = None if first_rows is None else (first_rows.data_ptr(), first_rows.numel())
        if key in cache and cache[key][0] == stamp:
            return cache[key][1]
        packed, k_pad = [], None
        with torch.no_grad():
            for i, m in enumerate(mods):
                if not isinstance(m, nn.Linear):
                    continue
                wt = m.weight.detach().float().t()  # (K, Nout)
                if not packed and first_rows is not None:
                    wt = wt[first_rows]
                nout = m.out_features
                npad = -(-nout // 16) * 16
                kk = wt.shape[0] if k_pad is None else k_pad  # (a padded predecessor hands over zero columns: zero rows here)
                full = wt.new_zeros((kk, npad))
                full[:wt.shape[0], :nout] = wt
                bias = None
                if m.bias is not None:
                    bias = wt.new_zeros(npad)
                    bias[:nout] = m.bias.detach().float()
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                packed.append((full.contiguous(), bias, relu, nout))
                k_pad = npad
        cache[key] = (stamp, packed)
        return packed

    def native_forward(self, x, first_rows=None):
        """x (..., K) -> (..., Nout_last) through v3d_linear_rows, one launch per Linear (+ bias + ReLU)."""
        from ..pointnet2.pointnet2_utils import linear_rows
        lead = x.shape[:-1]
        a = x.reshape(-1, x.shape[-1])
        if a.stride(1) != 1:
            a = a.contiguous()
        packed = self._packed(first_rows)
        for li, (w, bias, relu, nout) in enumerate(packed):
            # intermediate layers keep their padded width (zero columns: the next layer's zero rows); the last one stores nout columns
            a = linear_rows(a, w, bias, relu, n_store=nout if li == len(packed) - 1 else None)
        return a.reshape(*lead, a.shape[-1])

    def forward(self, x):
        if self.native_ok(x):
            return self.native_forward(x)
        return super().forward(x)

