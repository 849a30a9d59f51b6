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

    # ---- inference: the whole scale on csrc/sa_mlp.hip (no grouped tensor, no MIOpen 1x1 convolution)
    def _fusable(self, features):
        if self.training or torch.is_grad_enabled() or features is None or not features.is_cuda or self.pool_method != "max_pool":
            return False
        # the structural half depends on the module objects alone: judged once per set of children (an eager PV-RCNN frame asks six
        # times; the walk below was 15 us of host time each)
        key = tuple(id(m) for mlp in self.mlps for m in mlp) + tuple((g.nsample, g.use_xyz) for g in self.groupers)
        cache = self.__dict__.setdefault("_fusable_cache", {})
        if cache.get("key") != key:
            cache["key"], cache["ok"] = key, self._fusable_structure()
        return cache["ok"]

    def _fusable_structure(self):
        for g, mlp in zip(self.groupers, self.mlps):
            if not g.use_xyz or g.nsample not in (16, 32):
                return False
            mods = list(mlp)
            if not mods or any(not isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.ReLU)) for m in mods):
                return False
            # what v3d_sa_mlp_layer covers: 1x1 stride-1 convolutions, each followed by [BatchNorm2d] and a ReLU (relu is fixed
            # in the kernel), padded output widths it is instantiated for -- anything else takes the grouped torch path
            i = 0
            while i < len(mods):
                c = mods[i]
                if (not isinstance(c, nn.Conv2d) or c.kernel_size != (1, 1) or c.stride != (1, 1) or c.padding != (0, 0)
                        or c.groups != 1 or c.dilation != (1, 1) or (-(-c.out_channels // 16) * 16) not in self.FUSED_WIDTHS):
                    return False
                i += 1
                if i < len(mods) and isinstance(mods[i], nn.BatchNorm2d):
                    if mods[i].num_features != c.out_channels or not mods[i].track_running_stats:
                        return False
                    i += 1
                if i >= len(mods) or not isinstance(mods[i], nn.ReLU):
                    return False
                i += 1
        return True

    PAIR_BOTH_SCALES = True   # two scales of the same widths: one v3d_sa_mlp_pair2 launch (False: a v3d_sa_mlp_pair launch per scale)
    PAIR_FIRST_LAYERS = True  # two-layer scales: v3d_linear_rows on the database + v3d_sa_mlp_pair (False: a launch per layer)

    def _pair_pieces(self, packed):
        """([W1_k[4:]] side by side (Kf, sum N1_k), [W1_k[0:3] (3, N1_k)], column offsets) of the packed scales, cached with them."""
        cache = self.__dict__.setdefault("_pair_cache", {})
        firsts = tuple(ly[0][0] for ly in packed)
        if cache.get("key") is None or len(cache["key"]) != len(firsts) or any(a is not b for a, b in zip(cache["key"], firsts)):
            offs = [0]
            for w1 in firsts:
                offs.append(offs[-1] + w1.shape[1])
            cache.update(key=firsts, w1f=torch.cat([w1[4:] for w1 in firsts], dim=1).contiguous(),
                         wxs=[w1[0:3].contiguous() for w1 in firsts], offs=offs)
        return cache["w1f"], cache["wxs"], cache["offs"]

    FUSED_WIDTHS = (16, 32, 64, 96, 128, 192, 256)  # padded Nout of csrc/sa_mlp.hip:sa_mlp_layer_kernel<Nout/16>

    def _packed_layers(self, k):
        """[(W (K_in, Nout_pad), bias (Nout_pad))] of scale k: eval BatchNorm folded in, rows laid out for sa_mlp_layer (first
        layer: xyz rows 0-2, a zero row, then the feature rows; channel counts padded with zeros to multiples of 4 / 16),
        cached until a parameter or running statistic changes."""
        mods = list(self.mlps[k])
        lists = self.__dict__.setdefault("_pack_tensors", {})
        mkey = tuple(id(m) for m in mods)
        if k not in lists or lists[k][0] != mkey:  # (the parameter objects of these children, looked up once: nn.Module.__getattr__ is slow)
            convs = [m for m in mods if isinstance(m, nn.Conv2d)]
            bns = [m for m in mods if isinstance(m, nn.BatchNorm2d)]
            lists[k] = (mkey, convs, bns, [(c, "weight") for c in convs] + [(c, "bias") for c in convs if c.bias is not None] +
                        [(b, n) for b in bns for n in ("weight", "bias", "running_mean", "running_var")])
        _, convs, bns, names = lists[k]
        # (a Parameter may be REPLACED on its module -- .cuda(), load_state_dict(assign=True): read it through the module's dicts)
        stamp = tuple((t.data_ptr(), t._version) for t in (m._parameters.get(n, None) if n in m._parameters else m._buffers[n] for m, n in names))
        cache = self.__dict__.setdefault("_pack_cache", {})
        if k in cache and cache[k][0] == stamp:
            return cache[k][1]
        packed, k_prev_pad = [], None
        with torch.no_grad():
            for li, conv in enumerate(convs):
                w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).float()
                bias = conv.bias.detach().float() if conv.bias is not None else w.new_zeros(conv.out_channels)
                if bns:
                    bn = bns[li]
                    scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                    w, bias = w * scale[:, None], (bias - bn.running_mean.float()) * scale + bn.bias.float()
                wt = w.t().contiguous()  # (cin, cout)
                nout_pad = -(-conv.out_channels // 16) * 16
                if li == 0:
                    c = conv.in_channels - 3
                    kf = -(-c // 4) * 4
                    full = wt.new_zeros((4 + kf, nout_pad))
                    full[0:3, :conv.out_channels] = wt[0:3]
                    full[4:4 + c, :conv.out_channels] = wt[3:]
                else:
                    full = wt.new_zeros((k_prev_pad, nout_pad))
                    full[:conv.in_channels, :conv.out_channels] = wt
                bpad = bias.new_zeros(nout_pad)
                bpad[:conv.out_channels] = bias
                packed.append((full.contiguous(), bpad.contiguous()))
                k_prev_pad = nout_pad
        cache[k] = (stamp, packed)
        return packed

    def max_radius(self):
        return max(abs(float(g.radius)) for g in self.groupers)

    def out_channels(self):
        """Output channels per scale (the last convolution of every scale's MLP)."""
        key = tuple(id(m) for mlp in self.mlps for m in mlp)
        cache = self.__dict__.setdefault("_couts_cache", {})
        if cache.get("key") != key:
            cache["key"] = key
            cache["couts"] = [[mod for mod in mlp if isinstance(mod, nn.Conv2d)][-1].out_channels for mlp in self.mlps]
        return cache["couts"]

    def prep_features(self, features_pm):
        """(B, N, C) point-major features -> contiguous, channels padded with zeros to a multiple of 4 (what the kernels read)."""
        c = features_pm.shape[2]
        kf = -(-c // 4) * 4
        return features_pm.contiguous() if kf == c else torch.nn.functional.pad(features_pm, (0, kf - c)).contiguous()

    def pair_plan(self, feat_cols=None, width=None):
        """(W1 feature parts side by side (Kf, sum N1), [W1[0:3]], column offsets, packed layers) when every scale is a two-layer MLP
        the PAIR kernel covers, else None.  `feat_cols` / `width`: the caller's feature matrix has `width` columns and the module's
        channel j lives in column feat_cols[j] (PV-RCNN's raw-point source: the reflectance is column 3 of the (B, N, 4) cloud
        itself) -- the weight rows are laid out for THAT matrix (zero rows for the other columns: exact zeros in every sum), so that
        no padded copy of the features is made."""
        packed = [self._packed_layers(k) for k in range(len(self.groupers))]
        if not (self.PAIR_FIRST_LAYERS and all(len(ly) == 2 and ly[0][0].shape[1] <= 256 for ly in packed)):
            return None
        w1f, wxs, offs = self._pair_pieces(packed)
        if feat_cols is not None:
            cache = self.__dict__.setdefault("_pair_cols_cache", {})
            key = (tuple(feat_cols), int(width))
            if key not in cache or cache[key][0] is not w1f:
                wide = w1f.new_zeros((int(width), w1f.shape[1]))
                for j, col in enumerate(feat_cols):
                    wide[col] = w1f[j]
                cache[key] = (w1f, wide.contiguous())
            w1f = cache[key][1]
        return w1f, wxs, offs, packed

    def fused_forward(self, xyz, features_pm, new_xyz, out_pm=None, grid=None, neighbours=None, p_all=None, plan=False):
        """features_pm (B, N, C) POINT-major -> (B, M, sum(mlps[k][-1])) POINT-major: every scale's last layer writes its pooled rows
        straight into its column block (no torch.cat of the scales).  `out_pm`: a (B, M, >= that many) view with unit channel stride
        to write into -- a column block of the caller's keypoint feature matrix (detector/model.py point_feature_extract).  `grid`: the
        ball-query grid of `xyz` when the caller built it beforehand (PU.ball_query_grids: several databases in one launch);
        `neighbours` / `p_all`: the ball-query indices per scale / the first-layer products (B, N, sum N1) when the caller computed
        them with the other modules' in one launch each (PU.ball_query_pairs_many, PU.linear_rows_many on `prep_features`); `plan`:
        the caller's `pair_plan()` of this module (False: looked up here)."""
        b, n, c = features_pm.shape
        m = new_xyz.shape[1]
        kf = -(-c // 4) * 4
        # (with the first-layer products handed in, the features themselves are not read again: no padded copy is made)
        feat = features_pm if (p_all is not None and plan) else self.prep_features(features_pm)
        xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
        couts = self.out_channels()
        if out_pm is None:
            out_pm = torch.empty((b, m, sum(couts)), dtype=torch.float32, device=feat.device)
        if (out_pm.shape[:2] != (b, m) or out_pm.shape[2] < sum(couts) or out_pm.stride(2) != 1 or out_pm.stride(0) != m * out_pm.stride(1)):
            raise RuntimeError("fused_forward: out_pm must be a (B, M, >= C_out) view with unit channel stride and frames back to back")
        rows = out_pm.as_strided((b * m, out_pm.shape[2]), (out_pm.stride(1), 1), out_pm.storage_offset())
        if neighbours is None:
            if len(self.groupers) == 2:  # both scales in one pass over the database
                ga, gb = self.groupers
                neighbours = PU.ball_query_pair(ga.radius, ga.nsample, gb.radius, gb.nsample, xyz, new_xyz, grid=grid)
            else:
                neighbours = [PU.ball_query(g.radius, g.nsample, xyz, new_xyz, grid=grid) for g in self.groupers]
        col = 0
        if plan is False:
            plan = self.pair_plan()
        pair = plan is not None
        if pair:
            # the first layers' feature parts once per DATABASE point (N rows, not M * ns), all scales in ONE product (their weights
            # side by side); the rest of a first layer is rebuilt inside its second layer's launch (csrc/sa_mlp.hip PAIR)
            w1f, wxs, offs, packed = plan
            if p_all is None:
                p_all = PU.linear_rows(feat.reshape(b * n, kf), w1f).view(b, n, -1)
        else:
            packed = [self._packed_layers(k) for k in range(len(self.groupers))]
        if (pair and self.PAIR_BOTH_SCALES and len(self.groupers) == 2 and couts[0] == couts[1] and packed[0][0][0].shape[1] == packed[1][0][0].shape[1]
                and packed[0][1][0].shape == packed[1][1][0].shape and all(t.is_contiguous() for ly in packed for pr in ly for t in pr)):
            # both scales in ONE launch (the same widths: csrc/sa_mlp.hip v3d_sa_mlp_pair2)
            PU.sa_mlp_pair2(p_all[:, :, offs[0]:offs[1]], p_all[:, :, offs[1]:offs[2]], xyz, new_xyz, neighbours[0], neighbours[1],
                            wxs[0], packed[0][0][1], wxs[1], packed[1][0][1], packed[0][1][0], packed[0][1][1], packed[1][1][0],
                            packed[1][1][1], rows[:, 0:couts[0]], rows[:, couts[0]:2 * couts[0]], couts[0])
            return out_pm[:, :, :2 * couts[0]]
        for k, grouper in enumerate(self.groupers):
            layers = packed[k]
            ns = grouper.nsample
            idx = neighbours[k]
            last = dict(out=rows[:, col:col + couts[k]], n_store=couts[k])
            if pair:
                PU.sa_mlp_pair(p_all[:, :, offs[k]:offs[k + 1]], xyz, new_xyz, idx, wxs[k], layers[0][1], layers[1][0], layers[1][1],
                               True, True, **last)
                col += couts[k]
                continue
            x = PU.sa_mlp_layer(feat, layers[0][0], layers[0][1], True, len(layers) == 1, xyz=xyz, new_xyz=new_xyz, idx=idx,
                                **(last if len(layers) == 1 else {}))
            for li in range(1, len(layers)):
                x = PU.sa_mlp_layer(x, layers[li][0], layers[li][1], True, li == len(layers) - 1, groups=(b, m, ns),
                                    **(last if li == len(layers) - 1 else {}))
            col += couts[k]
        return out_pm[:, :, :col]

    def forward(self, xyz, features=None, new_xyz=None, features_pm=None):
        """xyz (B,N,3), features (B,C,N), new_xyz (B,M,3) -> (new_xyz, (B, sum(mlps[k][-1]), M)).  `features_pm` (B,N,C) may
        be given instead of `features` by callers that hold point-major features (saves two transposes)."""
        if new_xyz is None:
            idx = PU.furthest_point_sample(xyz, self.npoint)
            new_xyz = PU.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        if features is None and features_pm is not None and not self._fusable(features_pm):
            features = features_pm.transpose(1, 2).contiguous()
        if self._fusable(features if features is not None else features_pm):
            pm = features_pm if features_pm is not None else features.transpose(1, 2)
            return new_xyz, self.fused_forward(xyz, pm, new_xyz).transpose(1, 2)  # (a view: the kernels write point-major rows)
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            g = mlp(grouper(xyz, new_xyz, features))  # (B, C', M, ns)
            g = g.max(dim=3).values if self.pool_method == "max_pool" else g.mean(dim=3)
            outs.append(g)
        return new_xyz, torch.cat(outs, dim=1)
