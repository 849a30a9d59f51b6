"""CPU restatement of ONE SECOND train step (vision3d/train.py:58-66 over detector/second.py:20-30, detector/proposal.py:100-141).

TEST INFRASTRUCTURE ONLY (bench.py cpu_baseline of the train line; tests/test_oracle_selfcheck.py).  Voxelizer and rulebooks are
the scalar C oracle; the sparse convolutions are the gather -> GEMM -> scatter-add per kernel offset that spconv runs on a CPU
(sparse_cnn.py:15-30 through spconv's indice_conv), stated in differentiable torch CPU ops; BatchNorm in training mode, the dense
RPN / heads and the loss are the torch CPU ops the reference calls; clip_grad_norm_(35) and Adam as in train.py:66-67,101.
Takes a plain {name: numpy array} state dict with the reference's key layout -- this module does not import the product package.
"""
import math
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O
from .second_cpu import BN_EPS, SPMIDDLE_FHD, grid_shape, voxelize_batch


def _sparse_conv(feats, weight, nbr, n_out):
    """out[o] = sum_k feats[nbr[o, k]] @ weight[k] over the live pairs: gather, GEMM, scatter-add per offset."""
    k_vol = nbr.shape[1]
    w = weight.reshape(k_vol, weight.shape[-2], weight.shape[-1])
    out = feats.new_zeros((n_out, w.shape[-1]))
    for k in range(k_vol):
        col = nbr[:, k]
        rows = torch.nonzero(col >= 0).squeeze(1)
        if rows.numel():
            out = out.index_add(0, rows, feats.index_select(0, col[rows]) @ w[k])
    return out


def _focal(inputs, targets, alpha=0.25, gamma=2.0):
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    return (alpha * targets + (1 - alpha) * (1 - targets)) * ce * (1 - p_t) ** gamma


def train_step(sd, clouds, targets, voxel_size, bounds, n_cls=1, n_yaw=2, lam=2.0, max_pts=5, max_voxels=20000, optimizer=None):
    """One optimiser step on the CPU.  sd: {name: numpy} (copied into leaf tensors); targets: dict of numpy G_cls (B,n_cls,n_yaw,H,W),
    M_cls, G_reg (...,7), M_reg (...,1).  Returns (loss, seconds, params dict of torch tensors, optimizer) -- pass the last two
    back in to continue training."""
    t0 = time.perf_counter()
    if isinstance(next(iter(sd.values())), np.ndarray):
        params = {k: torch.tensor(v, requires_grad=v.dtype.kind == "f" and not k.endswith(("running_mean", "running_var")))
                  for k, v in sd.items()}
    else:
        params = sd
    leaves = [p for p in params.values() if p.requires_grad]
    if optimizer is None:
        optimizer = torch.optim.Adam(leaves, lr=0.01, betas=(0.9, 0.99), weight_decay=0.01)
    optimizer.zero_grad()
    vox, coords, occ = voxelize_batch(clouds, voxel_size, bounds, max_pts, max_voxels)
    feats = torch.from_numpy(O.vfe_mean(vox, occ))
    shape = grid_shape(bounds, voxel_size)
    books = {}
    for bi, block in enumerate(SPMIDDLE_FHD):
        for li, (kind, cin, cout, ks, st, pd, key) in enumerate(block):
            if kind == "subm":
                if key not in books:
                    books[key] = torch.from_numpy(O.subm_rulebook(coords, shape, ks).astype(np.int64))
                nbr = books[key]
            else:
                coords, nbr_np, shape = O.sparse_rulebook(coords, shape, ks, st, pd)
                nbr = torch.from_numpy(nbr_np.astype(np.int64))
            pre = f"cnn.blocks.{bi}.{li}"
            feats = _sparse_conv(feats, params[pre + ".0.weight"], nbr, nbr.shape[0])
            feats = F.relu(F.batch_norm(feats, params[pre + ".1.running_mean"], params[pre + ".1.running_var"],
                                        params[pre + ".1.weight"], params[pre + ".1.bias"], True, 0.01, BN_EPS))
    b = len(clouds)
    d, h, w_ = shape
    ct = torch.from_numpy(coords.astype(np.int64))
    dense = feats.new_zeros((b, d, h, w_, feats.shape[1])).index_put((ct[:, 0], ct[:, 1], ct[:, 2], ct[:, 3]), feats)
    x = dense.permute(0, 4, 1, 2, 3).reshape(b, feats.shape[1] * d, h, w_)

    def cbr(x, conv, bn, pad):
        x = F.conv2d(x, params[conv + ".weight"], None, padding=pad)
        return F.relu(F.batch_norm(x, params[bn + ".running_mean"], params[bn + ".running_var"], params[bn + ".weight"],
                                   params[bn + ".bias"], True, 0.01, BN_EPS))
    x = cbr(F.pad(x, (1, 1, 1, 1)), "rpn.down_block.1", "rpn.down_block.2", 0)
    for j in range(5):
        x = cbr(x, f"rpn.down_block.{4 + 3 * j}", f"rpn.down_block.{5 + 3 * j}", 1)
    x = cbr(x, "rpn.up_block.0", "rpn.up_block.1", 0)
    cls = F.conv2d(x, params["head.conv_cls.weight"], params["head.conv_cls.bias"])
    reg = F.conv2d(x, params["head.conv_reg.weight"], params["head.conv_reg.bias"])
    ny, nx = cls.shape[-2:]
    p_cls = cls.view(b, n_cls, n_yaw, ny, nx)
    p_reg = reg.view(b, n_cls, 7, n_yaw, ny, nx).permute(0, 1, 3, 4, 5, 2)
    g_cls, m_cls, g_reg, m_reg = (torch.from_numpy(np.asarray(targets[k])) for k in ("G_cls", "M_cls", "G_reg", "M_reg"))
    normalizer = m_reg.to(p_reg.dtype).sum().clamp_(min=1)
    cls_loss = (_focal(p_cls, g_cls.float()) * m_cls.to(p_cls.dtype)).sum() / normalizer
    per = F.smooth_l1_loss(p_reg, g_reg.float(), reduction="none")
    reg_loss = ((per[..., 0:3] + per[..., 3:6] + per[..., 6:7] / math.pi) * m_reg.to(p_reg.dtype)).sum() / normalizer
    loss = cls_loss + lam * reg_loss
    loss.backward()
    torch.nn.utils.clip_grad_norm_(leaves, max_norm=35)
    optimizer.step()
    return float(loss.detach()), time.perf_counter() - t0, params, optimizer
