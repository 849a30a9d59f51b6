"""SubMConv3d / SparseConv3d: rulebook (csrc/rulebook.hip) + fused forward (csrc/spconv.hip) + autograd.

Weight layout is spconv-1.x's (k0, k1, k2, Cin, Cout), so a reference-trained state_dict
(`cnn.blocks.{b}.{l}.0.weight`) loads unchanged.  Semantics: cross-correlation,
identical to nn.Conv3d with weight.permute(4, 3, 0, 1, 2) (tests/test_oracle_selfcheck.py, tests/test_gpu_conv3d_parity.py).

When gradients are required (training) the layer runs through spconv/functional.py (data gradient = the same
gather kernel on the transposed rulebook, weight gradient = deterministic MFMA reduction); inference keeps the
fused conv + folded-BatchNorm + ReLU launch.
"""
import math

import torch
from torch import nn

from .. import _lib as L
from .tensor import Rulebook, SparseConvTensor


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3
        return [int(x) for x in v]
    return [int(v)] * 3


def build_subm_rulebook(x, ksize):
    n = x.features.shape[0]
    k = ksize[0] * ksize[1] * ksize[2]
    dev = x.features.device
    nbr = torch.empty((k, max(n, 1)), dtype=torch.int32, device=dev)
    if n:
        lib = L.lib()
        ws = L.workspace(lib.v3d_rulebook_workspace(n, n, k), dev)
        with L.device_guard(dev):
            L.check(lib.v3d_rulebook_subm(L.ptr(x.indices), L.ptr(x.n_dev), n, L.host_i32(x.spatial_shape),
                                          L.host_i32(ksize), L.ptr(nbr), L.ptr(ws), ws.numel(), L.stream_ptr()),
                    "rulebook_subm")
    return Rulebook(nbr, max(n, 1), n, x.n_dev, x.indices, list(x.spatial_shape))


def build_sparse_rulebook(x, ksize, stride, padding):
    """Creates the output site list too.  One host read of the output count (as spconv does)."""
    n = x.features.shape[0]
    k = ksize[0] * ksize[1] * ksize[2]
    dev = x.features.device
    out_shape = [(x.spatial_shape[j] + 2 * padding[j] - ksize[j]) // stride[j] + 1 for j in range(3)]
    # every input reaches at most prod(ceil(k/s)) outputs; the grid itself bounds it as well
    fan = 1
    for j in range(3):
        fan *= -(-ksize[j] // stride[j])
    cap_out = max(1, min(n * fan, x.batch_size * out_shape[0] * out_shape[1] * out_shape[2]))
    coords_out = torch.empty((cap_out, 4), dtype=torch.int32, device=dev)
    nbr = torch.empty((k, cap_out), dtype=torch.int32, device=dev)
    n_out = torch.zeros((1,), dtype=torch.int32, device=dev)
    overflow = torch.zeros((1,), dtype=torch.int32, device=dev)
    if n:
        lib = L.lib()
        ws = L.workspace(lib.v3d_rulebook_workspace(n, cap_out, k), dev)
        with L.device_guard(dev):
            L.check(lib.v3d_rulebook_sparse(L.ptr(x.indices), L.ptr(x.n_dev), n, L.host_i32(x.spatial_shape),
                                            L.host_i32(ksize), L.host_i32(stride), L.host_i32(padding),
                                            L.ptr(coords_out), L.ptr(n_out), cap_out, L.ptr(nbr), L.ptr(overflow),
                                            L.ptr(ws), ws.numel(), L.stream_ptr()), "rulebook_sparse")
    else:
        nbr.fill_(-1)
    n_host, ovf = torch.stack((n_out, overflow)).flatten().tolist()
    if ovf > 0:
        raise RuntimeError("sparse rulebook overflow (internal capacity bound violated)")
    return Rulebook(nbr, cap_out, n_host, n_out, coords_out[:n_host], out_shape)


def sparse_conv_forward(features, weight, rb, scale=None, shift=None, relu=False, algo=0, packed=None, variant=0, precision="bf16x3"):
    """out (rb.n, Cout) = act((sum_k features[nbr[k]] @ weight[k]) * scale + shift).
    variant (algo 4 only; tests and benchmarks): 0 = the kernel is picked from the live row count; 1 / 5 / 10 force the 16-row,
    the 64-row LDS-shared-weights or the LDS-ring kernel (passed to the C ABI as a negative rows_hint).
    precision (algo 4 only): "bf16x3" or "fp32" (f16s: the input's scale entry is taken from the rows' own maximum by a small
    launch in front of the layer, so the op-by-op path needs no calibration and cannot leave the range); `packed` must be the
    image of that arithmetic (pack_sparse_weight(.., precision))."""
    feat = L.as_f32("sparse_conv", features)
    cin, cout = weight.shape[-2], weight.shape[-1]
    w = L.as_f32("sparse_conv", weight).reshape(-1, cin, cout)
    k = w.shape[0]
    if feat.shape[1] != cin or rb.nbr.shape[0] != k:
        raise RuntimeError("sparse_conv: weight/rulebook/feature shapes disagree")
    out = torch.empty((rb.n, cout), dtype=torch.float32, device=feat.device)
    if rb.n == 0:
        return out
    sc = None if scale is None else L.as_f32("sparse_conv", scale)
    sh = None if shift is None else L.as_f32("sparse_conv", shift)
    if algo == 0:  # default: bf16x3 row-owner kernel once the reduction dim fills an MFMA, fp32 wave kernel below
        algo = 4 if (cin >= 16 and cout % 16 == 0) else 3
    if algo == 4:  # bf16x3 row-owner kernel on pre-packed split weights (packed image cached per weight version)
        img = packed if packed is not None else pack_sparse_weight(w, k, cin, cout, precision)
        prec = L.PRECISIONS[precision]
        entry = None
        with L.device_guard(feat.device):
            if prec == L.PREC_F16S:
                entry = torch.empty(4, dtype=torch.float32, device=feat.device)
                L.check(L.lib().v3d_act_scale_from_rows(L.ptr(feat), None, feat.shape[0], cin, 0, L.ptr(entry),
                                                         L.ptr(L.scale_scratch(feat.device)), L.stream_ptr()), "act_scale_from_rows")
            L.check(L.lib().v3d_sparse_conv_fwd_packed(L.ptr(feat), L.ptr(img), L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap, k,
                                                        cin, cout, L.ptr(sc), L.ptr(sh), int(bool(relu)), L.ptr(out),
                                                        -int(variant) if variant else int(rb.n), prec, L.ptr(entry), None, None, None, None,
                                                        L.stream_ptr()), "sparse_conv_fwd_packed")
        return out
    with L.device_guard(feat.device):
        L.check(L.lib().v3d_sparse_conv_fwd(L.ptr(feat), L.ptr(w), L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap, k, cin,
                                            cout, L.ptr(sc), L.ptr(sh), int(bool(relu)), L.ptr(out), int(algo),
                                            L.stream_ptr()), "sparse_conv_fwd")
    return out


def pack_sparse_weight(w_flat, k, cin, cout, precision="bf16x3"):
    """(K,Cin,Cout) fp32 -> split 16-bit fragments in MFMA order (csrc/spconv.hip spconv_pack_weights_kernel) for the arithmetic."""
    lib = L.lib()
    img = torch.empty(int(lib.v3d_sparse_conv_weight_image_bytes(k, cin, cout)), dtype=torch.uint8, device=w_flat.device)
    with L.device_guard(w_flat.device):
        L.check(lib.v3d_sparse_conv_pack_weights(L.ptr(w_flat), k, cin, cout, L.PRECISIONS[precision], L.ptr(img), L.stream_ptr()),
                "sparse_conv_pack_weights")
    return img


class _SparseConvBase(nn.Module):
    subm = False
    # arithmetic of the INFERENCE forward (no autograd) of the packed kernel: "fp32" = f16s, the reference's fp32 spconv layers up to
    # summation noise; "bf16x3" = the scale-free 2^-17 product, which the autograd path always uses (gradients span too many binades
    # for one scale per tensor).  csrc/spconv.hip "the split-precision product".
    precision = "fp32"

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__()
        if _triple(dilation) != [1, 1, 1] or groups != 1:
            raise NotImplementedError("dilation/groups are not used by vision3d")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride = [1, 1, 1] if self.subm else _triple(stride)  # subm ignores stride (sparse_cnn.py:154 passes 3)
        self.padding = [k // 2 for k in self.kernel_size] if self.subm else _triple(padding)
        self.indice_key = indice_key
        self.algo = 0
        self.variant = 0  # tests / benchmarks: force a packed-kernel variant (see sparse_conv_forward)
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # nn.Conv-style init on the (k0,k1,k2,Cin,Cout) layout
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        bound = math.sqrt(6.0 / ((1 + 5) * fan_in))
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def _packed_weight(self, precision="bf16x3"):
        """Split/packed weight image, cached ON THE MODULE and refreshed when the parameter (or the arithmetic) changes."""
        w = self.weight
        stamp = (w.data_ptr(), w._version, str(w.device), precision)
        cache = self.__dict__.get("_pack_cache")
        if cache is None or cache[0] != stamp:
            k = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            flat = w.detach().to(torch.float32).reshape(k, self.in_channels, self.out_channels).contiguous()
            cache = (stamp, pack_sparse_weight(flat, k, self.in_channels, self.out_channels, precision))
            self.__dict__["_pack_cache"] = cache
        return cache[1]

    def rulebook(self, x):
        rb = x.indice_dict.get(("prebuilt", id(self)))  # training pre-pass (modules.prebuild_rulebooks)
        if rb is not None:
            return rb
        rb = x.find_indice_pair(self.indice_key)
        if rb is None:
            if self.subm:
                rb = build_subm_rulebook(x, self.kernel_size)
            else:
                rb = build_sparse_rulebook(x, self.kernel_size, self.stride, self.padding)
            if self.indice_key is not None:
                x.indice_dict[self.indice_key] = rb
        return rb

    def forward(self, x, scale=None, shift=None, relu=False):
        """`scale/shift/relu` are the fused epilogue used by SparseSequential for conv+BN(eval)+ReLU."""
        assert isinstance(x, SparseConvTensor)
        rb = self.rulebook(x)
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.features.requires_grad):
            from .functional import sparse_conv_autograd
            feats = sparse_conv_autograd(x.features, self.weight, rb, self.subm, self.algo)
            if self.bias is not None:
                feats = feats + self.bias
            if scale is not None:
                feats = feats * scale + shift
            if relu:
                feats = torch.relu(feats)
            out = SparseConvTensor(feats, rb.out_indices, rb.out_shape, x.batch_size)
            out.indice_dict = x.indice_dict
            out._n_dev = rb.n_dev
            return out
        if self.bias is not None:  # fold the bias into the affine epilogue
            b = self.bias.detach()
            shift = b if shift is None else shift + b * scale
            if scale is None:
                scale = torch.ones_like(b)
        packed = None
        if self.algo in (0, 4) and self.in_channels >= 16 and self.out_channels % 16 == 0:
            packed = self._packed_weight(self.precision)
        feats = sparse_conv_forward(x.features.detach(), self.weight.detach(), rb, scale, shift, relu, self.algo, packed, self.variant,
                                    self.precision)
        out = SparseConvTensor(feats, rb.out_indices, rb.out_shape, x.batch_size)
        out.indice_dict = x.indice_dict
        out._n_dev = rb.n_dev
        return out


class SubMConv3d(_SparseConvBase):
    subm = True


class SparseConv3d(_SparseConvBase):
    subm = False
