"""SparseSequential (spconv.SparseSequential): children named "0", "1", ...; sparse modules consume
the SparseConvTensor, plain nn.Modules are applied to `.features` (detector/sparse_cnn.py:15-30).

In eval mode the pattern [sparse conv, BatchNorm1d, ReLU] is executed as ONE kernel launch: the
batch-norm is folded to a per-channel scale/shift in the conv epilogue together with the ReLU.
"""
import torch
from torch import nn

from .conv import _SparseConvBase
from .tensor import SparseConvTensor


def fold_batchnorm(bn):
    """BatchNorm (eval) -> (scale, shift) with y = x * scale + shift."""
    inv = torch.rsqrt(bn.running_var + bn.eps)
    gamma = bn.weight if bn.weight is not None else torch.ones_like(inv)
    beta = bn.bias if bn.bias is not None else torch.zeros_like(inv)
    scale = (gamma * inv).detach()
    shift = (beta - bn.running_mean * gamma * inv).detach()
    return scale.contiguous(), shift.contiguous()


import os

# training-mode BatchNorm1d + ReLU on sparse features: csrc/sparse_bn.hip ("fused") or the torch modules ("torch")
FUSED_TRAINING_BN = os.environ.get("V3D_SPARSE_BN", "fused") == "fused"


import os


def prebuild_rulebooks(root, x):
    """Coordinate-only pre-pass for training: build every rulebook of the module tree `root` for the sparse tensor `x`
    BEFORE any convolution is enqueued, and hand them to the layers through `x.indice_dict` (keyed by module id).

    Rulebooks depend on coordinates only, but a strided rulebook ends with one host read of its output count.  Built
    lazily inside the layers, each of those reads waits for all the convolution / BatchNorm work enqueued before it and
    the host then has to refill the queue: with five reads per step the train step was host-bound (13.4 ms of GPU work
    in 17 ms).  In the pre-pass the reads only wait for the short rulebook kernels; afterwards the whole forward is
    enqueued without a single synchronisation."""
    from .conv import _SparseConvBase
    if os.environ.get("V3D_PREBUILD_RULEBOOKS", "1") == "0":  # A/B switch (bench notes in docs/rounds/design_rounds_1-4.md)
        return x
    cur = x
    for m in root.modules():  # registration order == execution order for (nested) Sequential trees
        if not isinstance(m, _SparseConvBase):
            continue
        rb = m.rulebook(cur)
        x.indice_dict[("prebuilt", id(m))] = rb
        if not m.subm:  # the next layers see the output sites; only coordinates matter here
            nxt = SparseConvTensor.__new__(SparseConvTensor)
            nxt.__dict__.update(dict(features=rb.out_indices.new_empty((rb.n, 0), dtype=torch.float32), indices=rb.out_indices,
                                     spatial_shape=list(rb.out_shape), batch_size=x.batch_size, indice_dict=x.indice_dict,
                                     _n_dev=rb.n_dev))
            cur = nxt
    return x


class SparseSequential(nn.Sequential):

    def _folded(self, bn):
        """Folded (scale, shift) of an eval-mode BatchNorm, cached until one of its tensors changes
        (tensor._version is bumped by every in-place update, load_state_dict included)."""
        cache = self.__dict__.setdefault("_fold_cache", {})
        tensors = (bn.running_mean, bn.running_var, bn.weight, bn.bias)
        stamp = tuple((t.data_ptr(), t._version) for t in tensors if t is not None) + (bn.eps,)
        hit = cache.get(id(bn))
        if hit is None or hit[0] != stamp:
            hit = (stamp,) + fold_batchnorm(bn)
            cache[id(bn)] = hit
        return hit[1], hit[2]

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, _SparseConvBase):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                nxt2 = mods[i + 2] if i + 2 < len(mods) else None
                if (isinstance(nxt, nn.BatchNorm1d) and not nxt.training and nxt.track_running_stats):
                    scale, shift = self._folded(nxt)
                    relu = isinstance(nxt2, nn.ReLU)
                    x = m(x, scale, shift, relu)
                    i += 3 if relu else 2
                    continue
                if (FUSED_TRAINING_BN and isinstance(nxt, nn.BatchNorm1d) and nxt.training and torch.is_grad_enabled()):
                    # training: conv (autograd through the HIP kernels) + fused batch-statistics BN (+ ReLU)
                    from .functional import sparse_batch_norm_relu, sparse_bn_supported
                    x = m(x)
                    if x.features.shape[0] > 0 and sparse_bn_supported(x.features, nxt):
                        relu = isinstance(nxt2, nn.ReLU)
                        x = x.replace_feature(sparse_batch_norm_relu(x.features, nxt, relu))
                        i += 3 if relu else 2
                        continue
                    i += 1
                    continue
                x = m(x)
            elif isinstance(m, SparseSequential):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.features.shape[0] > 0:
                    x = x.replace_feature(m(x.features))
            else:
                x = m(x)
            i += 1
        return x
