"""CPU restatement of PV-RCNN stage 2 (vision3d/detector/model.py:46-74, roi_grid_pool.py:51-72, refinement.py:47-50).

TEST INFRASTRUCTURE ONLY (bench.py cpu_baseline of the --mode pvrcnn line; the same composition tests/test_gpu_pointops.py checks
the GPU path against).  Index work -- farthest-point sampling, ball query, grouping -- is the scalar C oracle (pointnet2's
published loops); the shared MLPs, the BEV bilinear lookup and the reduction / refinement MLPs are the torch CPU modules handed in
by the caller (a CPU copy of the model's own sub-modules: plain nn.Conv2d / BatchNorm2d / Linear) -- this module imports nothing
from the product package.
"""
import time

import numpy as np
import torch

from . import oracle as O


def set_abstraction(sa_cpu, xyz, feat_cn, new_xyz):
    """PointnetSAModuleMSG by hand: per scale ball query + group (oracle), relative xyz in front of the features, the module's own
    shared MLP on the CPU, max over the samples; scales concatenated on the channels.  xyz (B, N, 3), feat_cn (B, C, N)."""
    outs = []
    for grouper, mlp in zip(sa_cpu.groupers, sa_cpu.mlps):
        idx = O.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
        g_xyz = O.group(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
        g = np.concatenate([g_xyz, O.group(np.ascontiguousarray(feat_cn), idx)], 1)
        with torch.no_grad():
            outs.append(mlp(torch.from_numpy(g)).max(3).values.numpy())
    return np.concatenate(outs, 1)


def stage2(cpu_model, points, cnn_features, bev_map, proposals, samples, num_keypoints):
    """points (B, N, 4), cnn_features [(xyz (B, n, 3), feat (B, n, C))], bev_map (B, C, H, W), proposals (B, n, 7), samples
    (B, n, m, 3) in [0, 1) -- all numpy -> (box_deltas, scores, seconds)."""
    t0 = time.perf_counter()
    xyz = np.ascontiguousarray(points[..., :3])
    picked = O.fps(xyz, num_keypoints)
    kp = np.stack([xyz[b][picked[b]] for b in range(xyz.shape[0])])
    sources = [(xyz, np.ascontiguousarray(points[..., 3:4]))] + list(cnn_features)
    pooled = [set_abstraction(pnet, np.ascontiguousarray(sx), np.ascontiguousarray(sf.transpose(0, 2, 1)), kp)
              for pnet, (sx, sf) in zip(cpu_model.pnets, sources)]
    with torch.no_grad():
        bev = cpu_model.bev(torch.from_numpy(bev_map), torch.from_numpy(kp)).numpy()
        pf = np.concatenate(pooled + [bev], 1)
        props = torch.from_numpy(proposals)
        b, n = props.shape[:2]
        pts = cpu_model.roi_grid_pool.sample_gridpoints(props, torch.from_numpy(samples)).reshape(b, -1, 3).numpy()
        m = samples.shape[2]
        sa = set_abstraction(cpu_model.roi_grid_pool.pnet, kp, pf, np.ascontiguousarray(pts))
        per_box = torch.from_numpy(sa).reshape(b, -1, n, m).permute(0, 2, 1, 3).reshape(b, n, -1)
        feats = cpu_model.roi_grid_pool.reduction(per_box)
        deltas, scores = cpu_model.refinement_layer(None, feats, props)
    return deltas.numpy(), scores.numpy(), time.perf_counter() - t0
