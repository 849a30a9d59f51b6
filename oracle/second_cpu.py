"""CPU restatement of the SECOND forward (vision3d/detector/second.py:20-35 and everything under it).

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).  Sparse parts run in
oracle/v3d_oracle.c (scalar C, one core); dense parts are the literal torch CPU ops the reference
calls (nn.Conv2d / BatchNorm2d / ReLU, detector/second.py:58-79; 1x1 heads, detector/proposal.py:19-22).
Input: a plain {name: numpy array} state dict with the reference's key layout -- this module does not
import the product package.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O

# (kind, cin, cout, ksize, stride, padding, indice_key) in execution order: sparse_cnn.py:151-175
SPMIDDLE_FHD = [
    [("subm", 4, 16, 3, 1, 1, "subm0"), ("subm", 16, 16, 3, 1, 1, "subm0"), ("sparse", 16, 32, 3, 2, 1, None)],
    [("subm", 32, 32, 3, 1, 1, "subm1"), ("subm", 32, 32, 3, 1, 1, "subm1"), ("sparse", 32, 64, 3, 2, 1, None)],
    [("subm", 64, 64, 3, 1, 1, "subm2"), ("subm", 64, 64, 3, 1, 1, "subm2"), ("subm", 64, 64, 3, 1, 1, "subm2"),
     ("sparse", 64, 64, 3, 2, [0, 1, 1], None)],
    [("subm", 64, 64, 3, 1, 1, "subm3"), ("subm", 64, 64, 3, 1, 1, "subm3"), ("subm", 64, 64, 3, 1, 1, "subm3"),
     ("sparse", 64, 64, [3, 1, 1], [2, 1, 1], 0, None)],
]
BN_EPS = 1e-3


def grid_shape(cfg_bounds, voxel_size):
    lower, upper = np.reshape(np.asarray(cfg_bounds, np.float64), (2, 3))
    shape = (upper - lower) / np.asarray(voxel_size, np.float64) + [0, 0, 1]
    return np.int32(shape)[::-1].tolist()


def voxelize_batch(clouds, voxel_size, bounds, max_pts, max_voxels):
    """core/preprocess.py:26-33: per-frame voxelisation, batch index prefixed, concatenated."""
    feats, coords, occ = [], [], []
    for b, cloud in enumerate(clouds):
        v, c, n = O.voxelize(cloud, voxel_size, bounds, max_pts, max_voxels)
        feats.append(v)
        coords.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
        occ.append(n)
    return np.concatenate(feats), np.concatenate(coords), np.concatenate(occ)


def bn_fold(sd, prefix):
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = (g / np.sqrt(v + np.float32(BN_EPS))).astype(np.float32)
    return scale, (b - m * scale).astype(np.float32)


def sparse_backbone(sd, feats, coords, shape, batch_size, timings=None):
    """14 sparse layers (conv -> BatchNorm1d eval -> ReLU) and .dense() -> (B, 128, H, W)."""
    books = {}
    layer_stats = []
    for bi, block in enumerate(SPMIDDLE_FHD):
        for li, (kind, cin, cout, ks, st, pd, key) in enumerate(block):
            t0 = time.perf_counter()
            if kind == "subm":
                if key not in books:
                    books[key] = O.subm_rulebook(coords, shape, ks)
                nbr, out_coords, out_shape = books[key], coords, shape
            else:
                out_coords, nbr, out_shape = O.sparse_rulebook(coords, shape, ks, st, pd)
            t1 = time.perf_counter()
            w = sd[f"cnn.blocks.{bi}.{li}.0.weight"]
            scale, shift = bn_fold(sd, f"cnn.blocks.{bi}.{li}.1")
            n_in = feats.shape[0]
            feats = O.sparse_conv_fwd(feats, w, nbr, scale, shift, relu=True)
            t2 = time.perf_counter()
            layer_stats.append(dict(name=f"{bi}.{li}", kind=kind, cin=cin, cout=cout, K=nbr.shape[1], n_in=n_in,
                                    n_out=nbr.shape[0], pairs=int((nbr >= 0).sum()), t_rulebook=t1 - t0, t_conv=t2 - t1))
            coords, shape = out_coords, out_shape
    dense = O.densify(feats, coords, batch_size, shape)
    b, c, d, h, w_ = dense.shape
    if timings is not None:
        timings["layers"] = layer_stats
    return dense.reshape(b, c * d, h, w_), feats, coords, layer_stats


def dense_rpn_head(sd, bev):
    """detector/second.py:58-94 + detector/proposal.py:94-97 with torch CPU ops."""
    T = lambda k: torch.from_numpy(np.ascontiguousarray(sd[k]))
    x = torch.from_numpy(bev)

    def cbr(x, conv, bn, pad):
        x = F.conv2d(x, T(conv + ".weight"), None, padding=pad)
        x = F.batch_norm(x, T(bn + ".running_mean"), T(bn + ".running_var"), T(bn + ".weight"), T(bn + ".bias"),
                         False, 0.0, BN_EPS)
        return F.relu(x)

    x = cbr(F.pad(x, (1, 1, 1, 1)), "rpn.down_block.1", "rpn.down_block.2", 0)
    for j in range(5):
        x = cbr(x, f"rpn.down_block.{4 + 3 * j}", f"rpn.down_block.{5 + 3 * j}", 1)
    x = cbr(x, "rpn.up_block.0", "rpn.up_block.1", 0)
    cls = F.conv2d(x, T("head.conv_cls.weight"), T("head.conv_cls.bias"))
    reg = F.conv2d(x, T("head.conv_reg.weight"), T("head.conv_reg.bias"))
    return x.numpy(), cls.numpy(), reg.numpy()


def second_forward(sd, clouds, voxel_size, bounds, max_pts=5, max_voxels=20000, timings=None):
    """Whole SECOND forward on the CPU.  Returns dict(voxel coords/mean, bev, rpn, cls, reg)."""
    t0 = time.perf_counter()
    vox, coords, occ = voxelize_batch(clouds, voxel_size, bounds, max_pts, max_voxels)
    mean = O.vfe_mean(vox, occ)
    t1 = time.perf_counter()
    shape = grid_shape(bounds, voxel_size)
    bev, last_feats, last_coords, stats = sparse_backbone(sd, mean, coords, shape, len(clouds), timings)
    t2 = time.perf_counter()
    rpn, cls, reg = dense_rpn_head(sd, bev)
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(voxelize=t1 - t0, sparse=t2 - t1, dense=t3 - t2)
    return dict(voxels=vox, coords=coords, occupancy=occ, mean=mean, bev=bev, rpn=rpn, cls=cls, reg=reg,
                layer_stats=stats)


# ---------------------------------------------------------------------------------------------------------------------------------
# The same forward in FLOAT64 from the definition (numpy matmuls over the C rulebooks, torch CPU double convolutions): not a
# restatement of what the reference computes -- the reference computes in fp32 -- but the yardstick for the STRICT elementwise bar
# of the fp32-class arithmetic: two fp32 pipelines differ from each other by twice their own summation noise (the fp32 restatement
# above against the GPU path: 2-4e-4 on entries above 1e-3 of the maximum), each differs from float64 by its own (<= 2e-4).
# ---------------------------------------------------------------------------------------------------------------------------------
def sparse_backbone64(sd, feats, coords, shape, batch_size):
    feats = np.asarray(feats, np.float64)
    books = {}
    for bi, block in enumerate(SPMIDDLE_FHD):
        for li, (kind, cin, cout, ks, st, pd, key) in enumerate(block):
            if kind == "subm":
                if key not in books:
                    books[key] = O.subm_rulebook(coords, shape, ks)
                nbr, out_coords, out_shape = books[key], coords, shape
            else:
                out_coords, nbr, out_shape = O.sparse_rulebook(coords, shape, ks, st, pd)
            w = np.asarray(sd[f"cnn.blocks.{bi}.{li}.0.weight"], np.float64).reshape(nbr.shape[1], cin, cout)
            pre = f"cnn.blocks.{bi}.{li}.1"
            g, b_, m, v = (np.asarray(sd[pre + k], np.float64) for k in (".weight", ".bias", ".running_mean", ".running_var"))
            scale = g / np.sqrt(v + BN_EPS)
            f = np.concatenate([feats, np.zeros((1, cin))], 0)  # row -1 = absent neighbour
            out = np.zeros((nbr.shape[0], cout))
            for k in range(nbr.shape[1]):
                out += f[nbr[:, k]] @ w[k]
            feats = np.maximum(out * scale + (b_ - m * scale), 0.0)
            coords, shape = out_coords, out_shape
    c = feats.shape[1]
    dense = np.zeros((batch_size, c, shape[0], shape[1], shape[2]))
    co = np.asarray(coords).reshape(-1, 4)
    dense[co[:, 0], :, co[:, 1], co[:, 2], co[:, 3]] = feats
    return dense.reshape(batch_size, c * shape[0], shape[1], shape[2])


def dense_rpn_head64(sd, bev):
    T = lambda k: torch.from_numpy(np.ascontiguousarray(sd[k])).double()
    x = torch.from_numpy(np.ascontiguousarray(bev)).double()

    def cbr(x, conv, bn, pad):
        x = F.conv2d(x, T(conv + ".weight"), None, padding=pad)
        x = F.batch_norm(x, T(bn + ".running_mean"), T(bn + ".running_var"), T(bn + ".weight"), T(bn + ".bias"), False, 0.0, BN_EPS)
        return F.relu(x)

    x = cbr(F.pad(x, (1, 1, 1, 1)), "rpn.down_block.1", "rpn.down_block.2", 0)
    for j in range(5):
        x = cbr(x, f"rpn.down_block.{4 + 3 * j}", f"rpn.down_block.{5 + 3 * j}", 1)
    x = cbr(x, "rpn.up_block.0", "rpn.up_block.1", 0)
    cls = F.conv2d(x, T("head.conv_cls.weight"), T("head.conv_cls.bias"))
    reg = F.conv2d(x, T("head.conv_reg.weight"), T("head.conv_reg.bias"))
    return x.numpy(), cls.numpy(), reg.numpy()


def second_forward64(sd, clouds, voxel_size, bounds, max_pts=5, max_voxels=20000, dense=True):
    """-> dict(bev, rpn, cls, reg) in float64 (rpn / cls / reg only with dense=True): the strict yardstick, see above."""
    vox, coords, occ = voxelize_batch(clouds, voxel_size, bounds, max_pts, max_voxels)
    mean = O.vfe_mean(vox, occ)
    bev = sparse_backbone64(sd, mean, coords, grid_shape(bounds, voxel_size), len(clouds))
    if not dense:
        return dict(bev=bev)
    rpn, cls, reg = dense_rpn_head64(sd, bev)
    return dict(bev=bev, rpn=rpn, cls=cls, reg=reg)


def proposals(cls, reg, anchors, n_cls, n_yaw, dof, topk, score_thresh):
    """detector/proposal.py:47-80 on the CPU: sigmoid, top-k, decode, batched rotated NMS (0.01), score cut."""
    b = cls.shape[0]
    ny, nx = cls.shape[-2:]
    cls_t = torch.from_numpy(cls).view(b, n_cls, n_yaw, ny, nx)
    reg_t = torch.from_numpy(reg).view(b, n_cls, dof, -1, ny, nx).permute(0, 1, 3, 4, 5, 2)
    scores, aidx = cls_t.sigmoid().reshape(b, n_cls, -1).topk(topk, -1)
    gi = aidx[..., None].expand(-1, -1, -1, dof)
    deltas = reg_t.reshape(b, n_cls, -1, dof).gather(2, gi)
    anc = torch.from_numpy(anchors).reshape(1, n_cls, -1, dof).expand(b, -1, -1, -1).gather(2, gi)
    diag = torch.linalg.vector_norm(anc[..., 3:5], dim=-1, keepdim=True)
    norm = torch.cat((diag, diag, anc[..., 5:6]), -1)
    boxes = torch.cat((deltas[..., :3] * norm + anc[..., :3], deltas[..., 3:6].exp() * anc[..., 3:6],
                       deltas[..., 6:] + anc[..., 6:]), -1).reshape(-1, dof)
    scores = scores.reshape(-1)
    bi = torch.arange(b).view(b, 1, 1).expand(b, n_cls, topk).reshape(-1)
    ci = torch.arange(n_cls).view(1, n_cls, 1).expand(b, n_cls, topk).reshape(-1)
    keep = O.batched_nms_rotated(boxes[:, [0, 1, 3, 4, 6]].numpy(), scores.numpy(), (ci + n_cls * bi).numpy(), 0.01)
    keep = torch.from_numpy(keep)
    boxes, bi, ci, scores = boxes[keep], bi[keep], ci[keep], scores[keep]
    m = scores > torch.tensor(score_thresh)[ci]
    return boxes[m].numpy(), bi[m].numpy(), ci[m].numpy(), scores[m].numpy()
