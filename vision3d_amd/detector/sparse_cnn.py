"""SECOND sparse 3-D backbones (interface of vision3d/detector/sparse_cnn.py:15-192) over vision3d_amd.spconv.

The networks are written as tables (`FHD_STAGES`, `FHD_LITE_STAGES`): one row per conv, read by `_build_layer`.  The
module tree -- and therefore every state_dict key -- is the reference's:
    blocks.{stage}.{layer}.0   sparse conv, weight (k, k, k, Cin, Cout), no bias
    blocks.{stage}.{layer}.1   BatchNorm1d(eps 1e-3, momentum 0.01)
    blocks.{stage}.{layer}.2   ReLU
(the lite variant has no per-stage nesting: blocks.{layer}.{0,1,2}).

    stage      shape (z, y, x)     stride
    input   [41, 1600, 1408]          1
    0       [21,  800,  704]          2
    1       [11,  400,  352]          4
    2       [ 5,  200,  176]          8
    3       [ 2,  200,  176]          8     -> .dense() -> BEV (B, 128, 200, 176)
"""
import numpy as np
import torch
from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm

from .. import spconv

BN_EPS, BN_MOMENTUM = 1e-3, 0.01


def _conv_bn_relu(conv):
    return spconv.SparseSequential(conv, nn.BatchNorm1d(conv.out_channels, eps=BN_EPS, momentum=BN_MOMENTUM), nn.ReLU())


def make_subm_layer(C_in, C_out, *args, **kwargs):
    """Submanifold 3x3x3 conv + BN + ReLU (sparse_cnn.py:15-21; extra positional args are accepted and, as upstream,
    land on the stride slot that a submanifold conv ignores)."""
    return _conv_bn_relu(spconv.SubMConv3d(C_in, C_out, 3, *args, bias=False, **kwargs))


def make_sparse_conv_layer(C_in, C_out, *args, **kwargs):
    """Strided sparse conv + BN + ReLU (sparse_cnn.py:24-30)."""
    return _conv_bn_relu(spconv.SparseConv3d(C_in, C_out, *args, bias=False, **kwargs))


# rows: ("subm", cout, indice_key) | ("down", cout, ksize, stride, padding); cin chains from the previous row
FHD_STAGES = (
    (("subm", 16, "subm0"), ("subm", 16, "subm0"), ("down", 32, 3, 2, 1)),
    (("subm", 32, "subm1"), ("subm", 32, "subm1"), ("down", 64, 3, 2, 1)),
    (("subm", 64, "subm2"), ("subm", 64, "subm2"), ("subm", 64, "subm2"), ("down", 64, 3, 2, [0, 1, 1])),
    (("subm", 64, "subm3"), ("subm", 64, "subm3"), ("subm", 64, "subm3"), ("down", 64, (3, 1, 1), (2, 1, 1), 0)),
)
FHD_LITE_STAGES = tuple((stage[-1],) for stage in FHD_STAGES)


def _build_layer(c_in, row):
    if row[0] == "subm":
        return make_subm_layer(c_in, row[1], 3, indice_key=row[2])
    _, c_out, ksize, stride, padding = row
    return make_sparse_conv_layer(c_in, c_out, ksize, stride, padding=padding)


def _build_stages(c_in, table):
    stages = []
    for rows in table:
        layers = []
        for row in rows:
            layers.append(_build_layer(c_in, row))
            c_in = row[1]
        stages.append(layers)
    return stages


def random_choice(x, n, dim=0, generator=None):
    """n rows of x drawn with replacement (numpy.random.choice stand-in, sparse_cnn.py:33-37)."""
    if dim != 0:
        raise NotImplementedError("random_choice draws along dim 0 only")
    return x[torch.randint(0, x.shape[0], (n,), device=x.device, generator=generator)]


def compute_grid_shape(cfg):
    """ZYX cell counts of the CNN grid: the voxelizer's grid with one extra cell in z (sparse_cnn.py:40-45,
    SURVEY.md H6)."""
    bounds = np.asarray(cfg.GRID_BOUNDS, dtype=np.float64).reshape(2, 3)
    cells_xyz = (bounds[1] - bounds[0]) / np.asarray(cfg.VOXEL_SIZE, dtype=np.float64) + np.array([0, 0, 1])
    return cells_xyz.astype(np.int32)[::-1].tolist()


class SparseCNNBase(nn.Module):
    """forward(features, coordinates, batch_size) -> ([(xyz, features)] * 4 padded per frame, BEV map)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.grid_shape = compute_grid_shape(cfg)
        # the reference keeps these as plain cuda attributes (H10); as buffers they follow .to()/.cuda()
        for name, values in (("base_voxel_size", cfg.VOXEL_SIZE), ("voxel_offset", cfg.GRID_BOUNDS[:3])):
            self.register_buffer(name, torch.tensor(values, dtype=torch.float32), persistent=False)
        self.pad_generator = None  # optional torch.Generator: reproducible padding draws (H12)
        self.make_blocks(cfg)

    def make_blocks(self, cfg):
        raise NotImplementedError("subclasses define self.blocks")

    def init_weights(self):
        """Reference initialisation (sparse_cnn.py:73-92): Kaiming fan-out on any Conv2d, unit BN."""
        for module in self.modules():
            if isinstance(module, nn.Conv2d):
                nn.init.kaiming_normal_(module.weight, a=0, mode="fan_out", nonlinearity="relu")
                if module.bias is not None:
                    nn.init.zeros_(module.bias)
            elif isinstance(module, _BatchNorm):
                nn.init.ones_(module.weight)
                nn.init.zeros_(module.bias)

    # ---- sparse levels -> padded point sets for the set-abstraction layers -------------------------------------

    def compute_pad_amounts(self, batch_index, batch_size):
        """Rows per frame and the shortfall of each frame against the fullest one (sparse_cnn.py:107-116; rows are
        frame-sorted, so frame boundaries come from one searchsorted)."""
        boundaries = torch.searchsorted(batch_index.contiguous(),
                                        torch.arange(batch_size + 1, device=batch_index.device, dtype=batch_index.dtype))
        per_frame = boundaries.diff()
        return per_frame.tolist(), (per_frame.max() - per_frame).tolist()

    def pad_batch(self, x, batch_index, batch_size):
        """Ragged per-frame rows -> (B, N_max, C); short frames are topped up with resampled rows of their own
        (sparse_cnn.py:118-126)."""
        if batch_size == 1:
            return x[None]
        per_frame, shortfall = self.compute_pad_amounts(batch_index, batch_size)
        frames = []
        for rows, extra in zip(x.split(per_frame), shortfall):
            frames.append(torch.cat((rows, random_choice(rows, extra, generator=self.pad_generator))))
        return torch.stack(frames)

    def to_global(self, stride, volume):
        """Active sites of one level -> (metric xyz of the cell corner, features), padded per frame
        (sparse_cnn.py:91-105).  indices are (b, z, y, x); flipped they read (x, y, z, b)."""
        ind = volume.indices
        if ind.is_cuda and ind.dtype == torch.int32 and ind.dim() == 2 and ind.shape[1] == 4 and ind.is_contiguous() and ind.data_ptr() % 16 == 0:
            # the statements below in one launch (csrc/pointops.hip v3d_voxel_centers): same values
            from .. import _lib as L
            consts = self.__dict__.setdefault("_global_consts", {})
            if stride not in consts:  # base_voxel_size * stride rounded in fp32 like the tensor product, once, on the host
                consts[stride] = ((self.base_voxel_size.detach().cpu().float() * stride).tolist(), self.voxel_offset.detach().cpu().float().tolist())
            (sx, sy, sz), (ox, oy, oz) = consts[stride]
            xyz = torch.empty((ind.shape[0], 3), dtype=torch.float32, device=ind.device)
            with L.device_guard(ind.device):
                L.check(L.lib().v3d_voxel_centers(L.ptr(ind), ind.shape[0], sx, sy, sz, ox, oy, oz, L.ptr(xyz), L.stream_ptr()), "voxel_centers")
            frame = ind[:, 0]
        else:
            xyzb = ind.flip(1)
            frame = xyzb[:, 3]
            xyz = xyzb[:, :3].float() * (self.base_voxel_size * stride) + self.voxel_offset
        return (self.pad_batch(xyz, frame, volume.batch_size),
                self.pad_batch(volume.features, frame, volume.batch_size))

    def to_global_torch(self, stride, volume):
        """The reference's statements op by op (the cross-check of the fused launch in the tests)."""
        xyzb = volume.indices.flip(1)
        frame = xyzb[:, 3]
        xyz = xyzb[:, :3].float() * (self.base_voxel_size * stride) + self.voxel_offset
        return (self.pad_batch(xyz, frame, volume.batch_size),
                self.pad_batch(volume.features, frame, volume.batch_size))

    def to_bev(self, volume):
        """(B, C, D, H, W) dense volume with z folded into the channels: (B, C * D, H, W)."""
        return volume.dense().flatten(1, 2)

    def forward(self, features, coordinates, batch_size):
        level = spconv.SparseConvTensor(features, coordinates.int(), self.grid_shape, batch_size)
        if self.training and torch.is_grad_enabled():
            spconv.prebuild_rulebooks(self.blocks, level)  # all host reads of the step happen here, before any conv
        points = []
        for stride, stage in zip(self.cfg.STRIDES, self.blocks):
            points.append(self.to_global(stride, level))
            level = stage(level)
        return points, self.to_bev(level)


class SpMiddleFHD(SparseCNNBase):

    def make_blocks(self, cfg):
        stages = _build_stages(cfg.C_IN, FHD_STAGES)
        self.blocks = spconv.SparseSequential(*(spconv.SparseSequential(*layers) for layers in stages))


class SpMiddleFHDLite(SparseCNNBase):

    def make_blocks(self, cfg):
        stages = _build_stages(cfg.C_IN, FHD_LITE_STAGES)
        self.blocks = spconv.SparseSequential(*(layers[0] for layers in stages))


CNN_FACTORY = {cls.__name__: cls for cls in (SpMiddleFHD, SpMiddleFHDLite)}
