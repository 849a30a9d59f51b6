"""The slice of the spconv-1.x API that vision3d calls, on MI355X.

    spconv.utils.VoxelGenerator(...).generate(points)        core/preprocess.py:18-23,30
    spconv.SparseConvTensor(features, indices, shape, B)     detector/second.py:42-44
        .features .indices .batch_size .spatial_shape .dense()   detector/sparse_cnn.py:99-104,130
    spconv.SubMConv3d / SparseConv3d / SparseSequential      detector/sparse_cnn.py:15-30,153-175

spconv itself (author's patched fork, unpinned) is absent from the reference tree, so behaviour
follows the published spconv-1.x semantics restated in oracle/v3d_oracle.c ("parity unpinned").
"""
from . import utils
from .tensor import SparseConvTensor
from .conv import SparseConv3d, SubMConv3d
from .modules import SparseSequential, prebuild_rulebooks

__all__ = ["utils", "SparseConvTensor", "SparseConv3d", "SubMConv3d", "SparseSequential"]
