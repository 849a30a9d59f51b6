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
    with torch.cuda.device(x.device):
        L.check(lib.v3d_sparse_conv_bwd_weight(L.ptr(x), L.ptr(g), L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap, k, cin, cout,
                                               L.ptr(dw), L.ptr(ws), ws.numel(), L.stream_ptr()), "sparse_conv_bwd_weight")
    return dw


def transpose_rulebook(rb, n_in, device):
    """Strided layers: (K, n_in) table of the output row each (input row, offset) feeds."""
    k = rb.nbr.shape[0]
    cap_in = max(n_in, 1)
    nbr_t = torch.empty((k, cap_in), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
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
