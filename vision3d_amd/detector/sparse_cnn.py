"""SECOND sparse 3-D backbones (interface of vision3d/detector/sparse_cnn.py:15-192) over
vision3d_amd.spconv.  Module tree (hence state_dict keys) is the reference's:
  blocks.{b}.{l}.0 = sparse conv (weight (k,k,k,Cin,Cout), no bias), .1 = BatchNorm1d(eps 1e-3,
  momentum 0.01), .2 = ReLU.

    block      shape (z, y, x)     stride
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


def make_subm_layer(C_in, C_out, *args, **kwargs):
    return spconv.SparseSequential(
        spconv.SubMConv3d(C_in, C_out, 3, *args, **kwargs, bias=False),
        nn.BatchNorm1d(C_out, eps=1e-3, momentum=0.01),
        nn.ReLU(),
    )


def make_sparse_conv_layer(C_in, C_out, *args, **kwargs):
    return spconv.SparseSequential(
        spconv.SparseConv3d(C_in, C_out, *args, **kwargs, bias=False),
        nn.BatchNorm1d(C_out, eps=1e-3, momentum=0.01),
        nn.ReLU(),
    )


def random_choice(x, n, dim=0, generator=None):
    assert dim == 0
    idx = torch.randint(0, x.size(0), (n,), device=x.device, generator=generator)
    return x[idx]


def compute_grid_shape(cfg):
    """ZYX grid of the CNN: voxelizer grid + 1 in z (sparse_cnn.py:40-45)."""
    lower, upper = np.reshape(cfg.GRID_BOUNDS, (2, 3))
    shape = (upper - lower) / np.r_[cfg.VOXEL_SIZE] + [0, 0, 1]
    return np.int32(shape)[::-1].tolist()


class SparseCNNBase(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.grid_shape = compute_grid_shape(cfg)
        # the reference keeps these as plain cuda attributes (H10); buffers follow .to()/.cuda()
        self.register_buffer("base_voxel_size", torch.tensor(cfg.VOXEL_SIZE, dtype=torch.float32), persistent=False)
        self.register_buffer("voxel_offset", torch.tensor(cfg.GRID_BOUNDS[:3], dtype=torch.float32), persistent=False)
        self.pad_generator = None  # optional torch.Generator for reproducible padding (H12)
        self.make_blocks(cfg)

    def make_blocks(self, cfg):
        raise NotImplementedError

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=0, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, _BatchNorm):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def compute_pad_amounts(self, batch_index, batch_size):
        edges = torch.arange(batch_size + 1, device=batch_index.device, dtype=batch_index.dtype)
        start = torch.searchsorted(batch_index.contiguous(), edges)  # torchsearchsorted (sparse_cnn.py:112)
        count = start[1:] - start[:-1]
        return count.tolist(), (count.max() - count).tolist()

    def pad_batch(self, x, batch_index, batch_size):
        """Ragged per-frame rows -> dense (B, N_max, C) by resampling (sparse_cnn.py:118-126)."""
        if batch_size == 1:
            return x.unsqueeze(0)
        count, pad = self.compute_pad_amounts(batch_index, batch_size)
        chunks = x.split(count)
        return torch.stack([torch.cat((c, random_choice(c, p, generator=self.pad_generator)))
                            for c, p in zip(chunks, pad)])

    def to_global(self, stride, volume):
        """Voxel indices -> metric xyz of the voxel corner, padded per frame (sparse_cnn.py:91-105)."""
        index = torch.flip(volume.indices, (1,))  # (x, y, z, b)
        xyz = index[..., 0:3].float() * (self.base_voxel_size * stride) + self.voxel_offset
        xyz = self.pad_batch(xyz, index[..., -1], volume.batch_size)
        feature = self.pad_batch(volume.features, index[..., -1], volume.batch_size)
        return xyz, feature

    def to_bev(self, volume):
        dense = volume.dense()
        N, C, D, H, W = dense.shape
        return dense.view(N, C * D, H, W)

    def forward(self, features, coordinates, batch_size):
        x0 = spconv.SparseConvTensor(features, coordinates.int(), self.grid_shape, batch_size)
        if self.training and torch.is_grad_enabled():
            spconv.prebuild_rulebooks(self.blocks, x0)  # all host reads of the step happen here, before any conv
        x1 = self.blocks[0](x0)
        x2 = self.blocks[1](x1)
        x3 = self.blocks[2](x2)
        x4 = self.to_bev(self.blocks[3](x3))
        levels = [self.to_global(s, v) for s, v in zip(self.cfg.STRIDES, (x0, x1, x2, x3))]
        return levels, x4


class SpMiddleFHD(SparseCNNBase):

    def make_blocks(self, cfg):
        self.blocks = spconv.SparseSequential(
            spconv.SparseSequential(
                make_subm_layer(cfg.C_IN, 16, 3, indice_key="subm0"),
                make_subm_layer(16, 16, 3, indice_key="subm0"),
                make_sparse_conv_layer(16, 32, 3, 2, padding=1),
            ),
            spconv.SparseSequential(
                make_subm_layer(32, 32, 3, indice_key="subm1"),
                make_subm_layer(32, 32, 3, indice_key="subm1"),
                make_sparse_conv_layer(32, 64, 3, 2, padding=1),
            ),
            spconv.SparseSequential(
                make_subm_layer(64, 64, 3, indice_key="subm2"),
                make_subm_layer(64, 64, 3, indice_key="subm2"),
                make_subm_layer(64, 64, 3, indice_key="subm2"),
                make_sparse_conv_layer(64, 64, 3, 2, padding=[0, 1, 1]),
            ),
            spconv.SparseSequential(
                make_subm_layer(64, 64, 3, indice_key="subm3"),
                make_subm_layer(64, 64, 3, indice_key="subm3"),
                make_subm_layer(64, 64, 3, indice_key="subm3"),
                make_sparse_conv_layer(64, 64, (3, 1, 1), (2, 1, 1)),
            ),
        )


class SpMiddleFHDLite(SparseCNNBase):

    def make_blocks(self, cfg):
        self.blocks = spconv.SparseSequential(
            make_sparse_conv_layer(cfg.C_IN, 32, 3, 2, padding=1),
            make_sparse_conv_layer(32, 64, 3, 2, padding=1),
            make_sparse_conv_layer(64, 64, 3, 2, padding=[0, 1, 1]),
            make_sparse_conv_layer(64, 64, (3, 1, 1), (2, 1, 1)),
        )


CNN_FACTORY = dict(SpMiddleFHD=SpMiddleFHD, SpMiddleFHDLite=SpMiddleFHDLite)
