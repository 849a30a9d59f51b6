"""PointnetSAModuleMSG as vision3d constructs it (detector/model.py:39-43, roi_grid_pool.py:28-32):
per scale ball-query+group -> shared MLP (Conv2d 1x1 no bias + BatchNorm2d + ReLU per layer) -> max
over the samples; scales concatenated on channels.  `mlps[i][0]` is incremented by 3 IN PLACE when
use_xyz (upstream behaviour -- the reason the reference deep-copies its config lists)."""
import torch
from torch import nn

from . import pointnet2_utils as PU


def shared_mlp(channels, bn=True):
    layers = []
    for cin, cout in zip(channels[:-1], channels[1:]):
        layers.append(nn.Conv2d(cin, cout, kernel_size=1, bias=not bn))
        if bn:
            layers.append(nn.BatchNorm2d(cout))
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class PointnetSAModuleMSG(nn.Module):

    def __init__(self, *, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, pool_method="max_pool"):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint, self.pool_method = npoint, pool_method
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(PU.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(shared_mlp(spec, bn=bn))

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N), new_xyz (B,M,3) -> (new_xyz, (B, sum(mlps[k][-1]), M))."""
        if new_xyz is None:
            idx = PU.furthest_point_sample(xyz, self.npoint)
            new_xyz = PU.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            g = mlp(grouper(xyz, new_xyz, features))  # (B, C', M, ns)
            g = g.max(dim=3).values if self.pool_method == "max_pool" else g.mean(dim=3)
            outs.append(g)
        return new_xyz, torch.cat(outs, dim=1)
