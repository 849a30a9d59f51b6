"""furthest_point_sample / gather_operation / ball_query / grouping_operation / QueryAndGroup
(csrc/pointops.hip).  Forward only in this round."""
import torch
from torch import nn

from .. import _lib as L


def furthest_point_sample(xyz, npoint):
    """xyz (B, N, 3) float32 -> idx (B, npoint) int32; first index 0, ties -> lowest index."""
    L.require_gpu("furthest_point_sample", xyz)
    p = L.as_f32("furthest_point_sample", xyz)
    b, n, _ = p.shape
    idx = torch.empty((b, npoint), dtype=torch.int32, device=p.device)
    with torch.cuda.device(p.device):
        L.check(L.lib().v3d_furthest_point_sample(L.ptr(p), b, n, int(npoint), L.ptr(idx), 0, 0, L.stream_ptr()),
                "furthest_point_sample")
    return idx


def gather_operation(features, idx):
    """features (B, C, N), idx (B, K) int32 -> (B, C, K)."""
    L.require_gpu("gather_operation", features, idx)
    f, i = L.as_f32("gather_operation", features), L.as_i32("gather_operation", idx)
    b, c, n = f.shape
    k = i.shape[1]
    out = torch.empty((b, c, k), dtype=torch.float32, device=f.device)
    with torch.cuda.device(f.device):
        L.check(L.lib().v3d_gather_points(L.ptr(f), L.ptr(i), b, c, n, k, L.ptr(out), L.stream_ptr()), "gather_points")
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz (B, N, 3), new_xyz (B, M, 3) -> idx (B, M, nsample) int32: the first `nsample` points (index
    order) with d^2 < r^2, empty slots filled with the first hit, no hit -> 0."""
    L.require_gpu("ball_query", xyz, new_xyz)
    p, q = L.as_f32("ball_query", xyz), L.as_f32("ball_query", new_xyz)
    b, n, _ = p.shape
    m = q.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=p.device)
    with torch.cuda.device(p.device):
        L.check(L.lib().v3d_ball_query(L.ptr(p), L.ptr(q), b, n, m, float(radius), int(nsample), L.ptr(idx),
                                       L.stream_ptr()), "ball_query")
    return idx


def grouping_operation(features, idx):
    """features (B, C, N), idx (B, M, ns) int32 -> (B, C, M, ns)."""
    L.require_gpu("grouping_operation", features, idx)
    f, i = L.as_f32("grouping_operation", features), L.as_i32("grouping_operation", idx)
    b, c, n = f.shape
    _, m, ns = i.shape
    out = torch.empty((b, c, m, ns), dtype=torch.float32, device=f.device)
    with torch.cuda.device(f.device):
        L.check(L.lib().v3d_group_points(L.ptr(f), L.ptr(i), b, c, n, m, ns, L.ptr(out), L.stream_ptr()),
                "group_points")
    return out


class QueryAndGroup(nn.Module):
    """Ball query + group; centres subtracted; xyz (3) concatenated BEFORE the features."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
