"""Sparse backbone vs torch.nn.functional.conv3d on the densified grid, stage by stage, on a FULL 16 384-point frame --
independent of oracle/ (VERDICT r2 item 7).

spconv's sources are absent from /root/reference, so the oracle's rulebook / sparse-convolution restatement is "parity
unpinned"; what IS pinned is the semantics the reference relies on (spconv/conv.py header): a
SubMConv3d / SparseConv3d equals nn.Conv3d with weight.permute(4, 3, 0, 1, 2) on the densified input, evaluated at the active
output sites, and a strided layer's output sites are exactly the positions whose receptive field holds an active input.
This test checks every one of the 14 layers of SpMiddleFHD (sparse_cnn.py:151-175) that way: the HIP layer's own input
(features + coordinates) is densified, F.conv3d (MIOpen / torch, fp32) is the reference, compared at the layer's output sites.
The grid is 41 x 1600 x 1408, so the dense reference runs in slabs of output rows with their halo.

Both arithmetics of the packed kernels (csrc/spconv.hip "the split-precision product"):
  "fp32"   (f16s, the default of the inference paths): the STRICT elementwise relative error against float64 on the entries with
           |ref| > 1e-3 max|ref| must stay below 2e-4 (torch's own fp32 conv3d shows up to 1.1e-4 under the same measure) and
           the feature bar is applied with its absolute floor cut to 1e-5 rms -- north_star's "within 1e-4 rel" read literally, up
           to fp32 summation noise;
  "bf16x3" (training plan, fast mode): the repository's bar of rounds 1-4 (1e-4 |ref| + 1e-4 rms_active, max norm), strict figure
           reported and bounded at what a 2^-17 product allows (3e-3).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import FP32_CLASS_FLOOR, STRICT_FP32_CLASS, assert_features_close

pytestmark = pytest.mark.gpu


def _dense_ref(feats, idx, shape, weight, stride, padding, out_idx, slab=160):
    """F.conv3d reference at `out_idx` (+ the reference's own set of active output sites), slab by slab along y."""
    dev = feats.device
    D, H, W = shape
    kz, ky, kx, cin, cout = weight.shape
    w = weight.permute(4, 3, 0, 1, 2).contiguous()
    ones = torch.ones((1, 1, kz, ky, kx), device=dev)
    Ho = (H + 2 * padding[1] - ky) // stride[1] + 1
    ref = torch.zeros((out_idx.shape[0], cout), dtype=torch.float32, device=dev)
    seen = torch.zeros(out_idx.shape[0], dtype=torch.bool, device=dev)
    n_ref_sites = 0
    for o0 in range(0, Ho, slab):
        o1 = min(Ho, o0 + slab)
        i0, i1 = o0 * stride[1] - padding[1], (o1 - 1) * stride[1] - padding[1] + ky  # input rows [i0, i1), may leave the grid
        sel = (idx[:, 2] >= i0) & (idx[:, 2] < i1)
        sub = idx[sel].long()
        dense = torch.zeros((1, cin, D, i1 - i0, W), dtype=torch.float32, device=dev)
        dense[0, :, sub[:, 1], sub[:, 2] - i0, sub[:, 3]] = feats[sel].t()
        occ = torch.zeros((1, 1, D, i1 - i0, W), dtype=torch.float32, device=dev)
        occ[0, 0, sub[:, 1], sub[:, 2] - i0, sub[:, 3]] = 1.0
        pad = (padding[0], 0, padding[2])  # the slab carries its own y halo (rows outside the grid are zero rows)
        out = F.conv3d(dense, w, None, stride, pad)          # (1, cout, Do, o1 - o0, Wo)
        cnt = F.conv3d(occ, ones, None, stride, pad)
        assert out.shape[3] == o1 - o0, (out.shape, o0, o1)
        n_ref_sites += int((cnt > 0.5).sum().item())
        osel = (out_idx[:, 2] >= o0) & (out_idx[:, 2] < o1)
        oi = out_idx[osel].long()
        ref[osel] = out[0, :, oi[:, 1], oi[:, 2] - o0, oi[:, 3]].t()
        assert bool((cnt[0, 0, oi[:, 1], oi[:, 2] - o0, oi[:, 3]] > 0.5).all()), "an output site with no active input in its field"
        seen |= osel
        del dense, occ, out, cnt
    assert bool(seen.all())
    return ref, n_ref_sites


def _exact64(feats, idx, shape, weight, stride, padding, out_idx):
    """The layer in float64, straight from the definition: out[o] = sum_k in[site(o * stride - padding + k)] @ W[k] over the active
    input sites (sorted linear keys + searchsorted: no rulebook of the library, no dense grid).  The yardstick that tells the
    kernel's error from the fp32 reference's own rounding."""
    D, H, W = shape
    kz, ky, kx, cin, cout = weight.shape
    key = lambda c: ((c[:, 0].long() * D + c[:, 1].long()) * H + c[:, 2].long()) * W + c[:, 3].long()
    kin, order = torch.sort(key(idx))
    f64, w64 = feats.double(), weight.double().reshape(kz * ky * kx, cin, cout)
    out = torch.zeros((out_idx.shape[0], cout), dtype=torch.float64, device=feats.device)
    o = out_idx.long()
    for a in range(kz):
        for b in range(ky):
            for c in range(kx):
                z, y, x = o[:, 1] * stride[0] - padding[0] + a, o[:, 2] * stride[1] - padding[1] + b, o[:, 3] * stride[2] - padding[2] + c
                ok = (z >= 0) & (z < D) & (y >= 0) & (y < H) & (x >= 0) & (x < W)
                want = ((o[:, 0] * D + z) * H + y) * W + x
                pos = torch.searchsorted(kin, want.clamp_min(0)).clamp_max(kin.numel() - 1)
                hit = ok & (kin[pos] == want)
                rows = torch.nonzero(hit).squeeze(1)
                if rows.numel():
                    out.index_add_(0, rows, f64[order[pos[rows]]] @ w64[(a * ky + b) * kx + c])
    return out


def _strict(got, ref):
    """largest elementwise relative error over the entries with |ref| > 1e-3 max|ref|"""
    big = ref.abs() > 1e-3 * ref.abs().max()
    return float(((got.double() - ref.double()).abs()[big] / ref.double().abs()[big]).max())


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_every_backbone_layer_equals_dense_conv3d_on_a_full_frame(precision, monkeypatch):
    from vision3d_amd import spconv, synth
    from vision3d_amd.core import Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import Second
    from vision3d_amd.spconv.conv import _SparseConvBase
    monkeypatch.setattr(_SparseConvBase, "precision", precision)
    fp32_class = precision == "fp32"
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = Second(cfg).cuda().eval()
    cloud = torch.from_numpy(synth.make_cloud(11, 16384)).cuda()
    captured = []

    def hook(mod, args, out):
        x = args[0]
        captured.append((mod, x.features.detach().clone(), x.indices.clone(), list(x.spatial_shape), args[1:],
                         out.features.detach().clone(), out.indices.clone(), list(out.spatial_shape)))
    hooks = [m.register_forward_hook(hook) for m in model.cnn.modules() if isinstance(m, _SparseConvBase)]
    with torch.no_grad():
        it = Preprocessor(cfg, seed=0)(dict(points=[cloud]))
        x = spconv.SparseConvTensor(it["voxel_mean"], it["coordinates"].int(), model.cnn.grid_shape, 1)
        model.cnn.blocks(x)
    for h in hooks:
        h.remove()
    assert len(captured) == 14, len(captured)
    assert captured[0][1].shape[0] > 10000  # a full frame, not a crop
    torch.backends.cudnn.allow_tf32 = False
    worst_strict, worst_vs_exact, worst_torch, report = 0.0, 0.0, 0.0, []
    for li, (mod, fin, iin, shp, extra, fout, iout, oshp) in enumerate(captured):
        n_out = fout.shape[0]
        scale, shift, relu = (list(extra) + [None, None, False])[:3]
        with torch.no_grad():
            ref, n_sites = _dense_ref(fin, iin, shp, mod.weight.detach().float(), mod.stride, mod.padding, iout[:n_out])
            if mod.bias is not None:
                ref = ref + mod.bias
            if scale is not None:
                ref = ref * scale + shift
            if relu:
                ref = torch.relu(ref)
        if mod.subm:
            assert torch.equal(iin[:n_out], iout[:n_out])  # submanifold: the output sites ARE the input sites
        else:
            assert n_sites == n_out, f"layer {li}: {n_out} output sites, dense occupancy conv has {n_sites}"
            assert len(torch.unique(iout[:n_out], dim=0)) == n_out
            assert [int(v) for v in oshp] == [(shp[j] + 2 * mod.padding[j] - mod.kernel_size[j]) // mod.stride[j] + 1 for j in range(3)]
        with torch.no_grad():  # the same layer in float64 from the definition: whose rounding is the strict figure?
            exact = _exact64(fin, iin, shp, mod.weight.detach(), mod.stride, mod.padding, iout[:n_out])
            if mod.bias is not None:
                exact = exact + mod.bias.double()
            if scale is not None:
                exact = exact * scale.double() + shift.double()
            if relu:
                exact = torch.relu(exact)
        kernel_vs_exact, torch_vs_exact = _strict(fout, exact), _strict(ref, exact)
        worst_vs_exact, worst_torch = max(worst_vs_exact, kernel_vs_exact), max(worst_torch, torch_vs_exact)
        got, r = fout.cpu().numpy(), ref.cpu().numpy()
        assert_features_close(got, r, f"layer {li} {mod.in_channels}->{mod.out_channels} vs F.conv3d",
                              floor=FP32_CLASS_FLOOR if fp32_class else 1e-4)
        big = np.abs(r) > 1e-3 * np.abs(r).max()
        strict = float((np.abs(got - r)[big] / np.abs(r)[big]).max())
        worst_strict = max(worst_strict, strict)
        report.append((li, mod.in_channels, mod.out_channels, n_out, strict, float(np.abs(got - r).max() / np.abs(r).max()),
                       kernel_vs_exact, torch_vs_exact))
    for row in report:
        print("layer %2d %3d->%3d rows %6d: strict rel err on |ref| > 1e-3 max = %.2e, max-norm err = %.2e | vs float64: kernel %.2e, "
              "torch fp32 conv3d %.2e" % row)
    # Against the float64 result: the kernel's strict error, with the fp32 reference's own figure printed beside it (two fp32-class
    # computations with different summation orders differ by that much on small entries).
    print("%s: worst strict error vs float64: kernel %.2e, torch fp32 conv3d %.2e; kernel vs torch %.2e" %
          (precision, worst_vs_exact, worst_torch, worst_strict))
    if fp32_class:
        assert worst_vs_exact < STRICT_FP32_CLASS, worst_vs_exact
        assert worst_strict < 2 * STRICT_FP32_CLASS, worst_strict  # (two fp32-class results against each other)
    else:
        # the bf16x3 product carries an ABSOLUTE error of a few 1e-6 of the layer's largest output, so relative to an entry a
        # thousand times smaller it reaches a few 1e-3 (observed 1.2e-3 .. 2.0e-3 over the 14 layers)
        assert worst_vs_exact < 3e-3, worst_vs_exact
        assert worst_strict < 3e-3, worst_strict
