"""Autograd for the sparse convolution (training path, reference: train.py:63-67 through spconv).

forward   out[o] = sum_k x[nbr[k][o]] @ W[k]                                  (csrc/spconv.hip)
backward  dX[i]  = sum_k dY[nbrT[k][i]] @ W[k]^T   -- the SAME gather kernel on the transposed rulebook:
                     submanifold: nbrT[k] = nbr[K-1-k]  (o = i + off_k  <=>  i = o + off_{K-1-k})
                     strided:     nbrT built by v3d_rulebook_transpose
          dW[k]  = sum_{pairs} x[i]^T dY[o]         -- exact-fp32 MFMA reduction, deterministic
"""
import torch

from .. import _lib as L
from .tensor import Rulebook


def sparse_conv_bwd_weight(features, grad_out, rb, k, cin, cout):
    lib = L.lib()
    x, g = L.as_f32("bwd_weight", features), L.as_f32("bwd_weight", grad_out)
    dw = torch.empty((k, cin, cout), dtype=torch.float32, device=x.device)
    if rb.n == 0:
        return dw.zero_()
    ws = L.workspace(lib.v3d_sparse_conv_bwd_weight_workspace(k, cin, cout), x.device)
    with L.device_guard(x.device):
        L.check(lib.v3d_sparse_conv_bwd_weight(L.ptr(x), L.ptr(g), L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap, k, cin, cout,
                                               L.ptr(dw), L.ptr(ws), ws.numel(), L.stream_ptr()), "sparse_conv_bwd_weight")
    return dw


def transpose_rulebook(rb, n_in, device):
    """Strided layers: (K, n_in) table of the output row each (input row, offset) feeds."""
    k = rb.nbr.shape[0]
    cap_in = max(n_in, 1)
    nbr_t = torch.empty((k, cap_in), dtype=torch.int32, device=device)
    with L.device_guard(device):
        L.check(L.lib().v3d_rulebook_transpose(L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap, k, cap_in, L.ptr(nbr_t), L.stream_ptr()),
                "rulebook_transpose")
    n_dev = torch.tensor([n_in], dtype=torch.int32, device=device)
    return Rulebook(nbr_t, cap_in, n_in, n_dev, None, None)


class SparseConvFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, features, weight, rb, subm, algo):
        from .conv import sparse_conv_forward
        ctx.save_for_backward(features, weight)
        ctx.rb, ctx.subm, ctx.algo = rb, subm, algo
        return sparse_conv_forward(features.detach(), weight.detach(), rb, None, None, False, algo)

    @staticmethod
    def backward(ctx, grad_out):
        from .conv import sparse_conv_forward
        features, weight = ctx.saved_tensors
        rb = ctx.rb
        cin, cout = weight.shape[-2], weight.shape[-1]
        w = weight.detach().reshape(-1, cin, cout)
        k = w.shape[0]
        g = grad_out.contiguous()
        grad_f = grad_w = None
        if ctx.needs_input_grad[1]:
            grad_w = sparse_conv_bwd_weight(features.detach(), g, rb, k, cin, cout).view_as(weight)
        if ctx.needs_input_grad[0]:
            n_in = features.shape[0]
            if ctx.subm:
                rb_t = rb                                               # its own transpose ...
                w_t = w.flip(0).transpose(1, 2).contiguous()            # ... with the offsets reversed
            else:
                rb_t = transpose_rulebook(rb, n_in, features.device)
                w_t = w.transpose(1, 2).contiguous()
            algo = 0 if cin % 16 == 0 else 1  # the transposed layer has Cout' = Cin: MFMA kernels need a multiple of 16
            grad_f = sparse_conv_forward(g, w_t, rb_t, None, None, False, algo)
        return grad_f, grad_w, None, None, None


def sparse_conv_autograd(features, weight, rb, subm, algo=0):
    return SparseConvFunction.apply(features, weight, rb, subm, algo)


_SBN_WS = {}


def _sbn_workspace(dev):
    """One scratch buffer per device, sized for the largest supported C (512 chunks x 3 x 256 floats): the ops run in
    stream order, so consecutive layers can share it; no allocator traffic per call."""
    ws = _SBN_WS.get(dev)
    if ws is None:
        ws = _SBN_WS[dev] = L.workspace(L.lib().v3d_sparse_bn_workspace(1, 256), dev)
    return ws


class SparseBatchNormReLUFunction(torch.autograd.Function):
    """Training-mode BatchNorm1d (+ ReLU) on sparse features through csrc/sparse_bn.hip (3 launches each way, the
    running statistics are updated by the merge kernel)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu, running_mean, running_var, momentum, num_batches_tracked):
        lib = L.lib()
        xf = x if (x.dtype == torch.float32 and x.is_contiguous()) else L.as_f32("sparse_bn", x)
        n, c = xf.shape
        dev = xf.device
        y = torch.empty_like(xf)
        stats = torch.empty((3, c), dtype=torch.float32, device=dev)  # save_mean | save_invstd | unbiased variance
        ws = _sbn_workspace(dev)
        with L.device_guard(dev):
            L.check(lib.v3d_sparse_bn_relu_fwd(xf.data_ptr(), n, c, weight.data_ptr(), bias.data_ptr(), float(eps), int(bool(relu)),
                                               y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                                               L.ptr(running_mean), L.ptr(running_var), float(momentum),
                                               L.ptr(num_batches_tracked), ws.data_ptr(), ws.numel(), L.stream_ptr()),
                    "sparse_bn_relu_fwd")
        ctx.save_for_backward(xf, weight, bias, stats)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        x, weight, bias, stats = ctx.saved_tensors
        n, c = x.shape
        dev = x.device
        g = dy if (dy.dtype == torch.float32 and dy.is_contiguous()) else L.as_f32("sparse_bn", dy)
        dx = torch.empty_like(x)
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        ws = _sbn_workspace(dev)
        with L.device_guard(dev):
            L.check(lib.v3d_sparse_bn_relu_bwd(x.data_ptr(), g.data_ptr(), n, c, weight.data_ptr(), bias.data_ptr(),
                                               stats[0].data_ptr(), stats[1].data_ptr(), int(ctx.relu), dx.data_ptr(),
                                               dgb[0].data_ptr(), dgb[1].data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr()),
                    "sparse_bn_relu_bwd")
        return dx, dgb[0], dgb[1], None, None, None, None, None, None


def sparse_batch_norm_relu(features, bn, relu):
    """nn.BatchNorm1d in TRAINING mode (+ optional ReLU) on (n, C) features; the running statistics are updated exactly
    as the module would (momentum, unbiased variance, num_batches_tracked) -- inside the merge kernel."""
    track = bn.track_running_stats and bn.running_mean is not None
    if track and bn.momentum is None:  # cumulative moving average: 1 / (batches seen, including this one)
        momentum = 1.0 / float(int(bn.num_batches_tracked) + 1)
    else:
        momentum = bn.momentum if bn.momentum is not None else 0.0
    return SparseBatchNormReLUFunction.apply(features, bn.weight, bn.bias, bn.eps, relu,
                                             bn.running_mean if track else None, bn.running_var if track else None, momentum,
                                             bn.num_batches_tracked if track else None)


def sparse_bn_supported(features, bn):
    n, c = features.shape
    return features.is_cuda and n > 1 and 4 <= c <= 256 and (c & (c - 1)) == 0 and features.dtype == torch.float32 and bn.affine
