"""GPU parity: hash rulebooks (bit-exact vs oracle/) and sparse convolution forward (both kernels, every
SECOND layer shape) within 1e-4 relative of oracle/, plus .dense() exact."""
import numpy as np
import pytest
import torch

from gpu_util import FP32_CLASS_FLOOR, STRICT_FP32_CLASS, assert_features_close, dev, strict_rel_err
from vision3d_amd import synth

pytestmark = pytest.mark.gpu


def conv64(feats, w, nbr, scale=None, shift=None, relu=False):
    """The layer in float64 from the oracle's neighbour table (n_out, K): the yardstick of the fp32-class arithmetic."""
    k = nbr.shape[1]
    w64 = w.reshape(k, w.shape[-2], w.shape[-1]).astype(np.float64)
    f64 = np.concatenate([feats.astype(np.float64), np.zeros((1, feats.shape[1]))], 0)  # row -1 = absent neighbour
    out = np.zeros((nbr.shape[0], w.shape[-1]))
    for j in range(k):
        out += f64[nbr[:, j]] @ w64[j]
    if scale is not None:
        out = out * scale.astype(np.float64) + shift.astype(np.float64)
    return np.maximum(out, 0) if relu else out


def assert_fp32_class(got, ref64, what):
    assert_features_close(got, ref64, what, floor=FP32_CLASS_FLOOR)
    assert strict_rel_err(got, ref64) < STRICT_FP32_CLASS, (what, strict_rel_err(got, ref64))


def kitti_coords(oracle, seeds):
    from oracle import second_cpu
    clouds = [synth.make_cloud(s) for s in seeds]
    _, coords, _ = second_cpu.voxelize_batch(clouds, [0.05, 0.05, 0.1], synth.KITTI_BOUNDS, 5, 20000)
    return coords


def make_tensor(coords, feats, shape, b):
    from vision3d_amd.spconv import SparseConvTensor
    return SparseConvTensor(dev(feats, torch.float32), dev(coords, torch.int32), shape, b)


def test_rulebooks_bit_exact_through_the_backbone(oracle):
    """subm + strided rulebooks for all four stage transitions of SpMiddleFHD, batch of 2."""
    from vision3d_amd.spconv.conv import build_sparse_rulebook, build_subm_rulebook
    coords = kitti_coords(oracle, [0, 1])
    shape = [41, 1600, 1408]
    specs = [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1]),
             ([3, 1, 1], [2, 1, 1], [0, 0, 0])]
    for ks, st, pd in specs:
        x = make_tensor(coords, np.zeros((len(coords), 1), np.float32), shape, 2)
        rb = build_subm_rulebook(x, [3, 3, 3])
        np.testing.assert_array_equal(rb.nbr.cpu().numpy().T, oracle.subm_rulebook(coords, shape, 3))
        rs = build_sparse_rulebook(x, ks, st, pd)
        oc, onbr, oshape = oracle.sparse_rulebook(coords, shape, ks, st, pd)
        assert rs.n == len(oc) and rs.out_shape == oshape
        np.testing.assert_array_equal(rs.out_indices.cpu().numpy(), oc)
        np.testing.assert_array_equal(rs.nbr[:, :rs.n].cpu().numpy().T, onbr)
        coords, shape = oc, oshape
    assert shape == [2, 200, 176]


@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)])
@pytest.mark.parametrize("algo", [1, 3, 4])
def test_subm_conv_forward(oracle, cin, cout, algo):
    from vision3d_amd.spconv.conv import build_subm_rulebook, sparse_conv_forward
    rng = np.random.default_rng(cin * 100 + cout)
    coords = kitti_coords(oracle, [2])[:6000]
    feats = rng.standard_normal((len(coords), cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32) * 0.1
    x = make_tensor(coords, feats, [41, 1600, 1408], 1)
    rb = build_subm_rulebook(x, [3, 3, 3])
    nbr = oracle.subm_rulebook(coords, [41, 1600, 1408], 3)
    got = sparse_conv_forward(x.features, dev(w), rb, dev(sc), dev(sh), True, algo).cpu().numpy()
    assert_features_close(got, oracle.sparse_conv_fwd(feats, w, nbr, sc, sh, True), f"subm {cin}->{cout} algo {algo} fused")
    got = sparse_conv_forward(x.features, dev(w), rb, None, None, False, algo).cpu().numpy()
    assert_features_close(got, oracle.sparse_conv_fwd(feats, w, nbr), f"subm {cin}->{cout} algo {algo} plain")


@pytest.mark.parametrize("algo", [1, 3, 4])
def test_strided_conv_forward_ragged_tail(oracle, algo):
    """Strided layers incl. the (3,1,1) one; row counts not multiples of the 64-row tile; batch 2."""
    from vision3d_amd.spconv.conv import build_sparse_rulebook, sparse_conv_forward
    rng = np.random.default_rng(9)
    coords = kitti_coords(oracle, [3, 4])[:7001]
    shape = [41, 1600, 1408]
    for (cin, cout, ks, st, pd) in [(16, 32, [3, 3, 3], [2, 2, 2], [1, 1, 1]), (64, 64, [3, 1, 1], [2, 1, 1], [0, 0, 0])]:
        feats = rng.standard_normal((len(coords), cin)).astype(np.float32)
        w = (rng.standard_normal((*ks, cin, cout)) / np.sqrt(cin * 3)).astype(np.float32)
        x = make_tensor(coords, feats, shape, 2)
        rb = build_sparse_rulebook(x, ks, st, pd)
        oc, onbr, _ = oracle.sparse_rulebook(coords, shape, ks, st, pd)
        got = sparse_conv_forward(x.features, dev(w), rb, None, None, True, algo).cpu().numpy()
        assert_features_close(got, oracle.sparse_conv_fwd(feats, w, onbr, relu=True), f"strided {cin}->{cout} algo {algo}")


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("variant", [1, 5, 6, 7, 10, 16],
                         ids=["16rows", "64rows_lds_weights", "offset_outer_staged", "offset_outer_regs", "lds_ring", "lds_ring_regs"])
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 32), (64, 64)])
def test_packed_kernel_variants(oracle, cin, cout, variant, precision):
    """Every kernel of the packed (algo 4) product -- the 16-row kernel, the two-tile LDS-ring kernel (3x3x3, the default
    up to 16 k rows: wave-specialised weight movers; in the form chosen per shape -- rows staged through LDS by
    row-contiguous LDS-DMA for 64->64 and 32->32 -- and in its register-gather form), the 64-row LDS-shared-weights kernel
    and the offset-outer persistent kernel (the 64->64 default from 32 k rows; staged and register forms) -- forced on a
    small problem through the per-call variant argument (a negative rows_hint at the C ABI; the library has no global
    switch): same result as the oracle, ragged tail (5003 rows: a half-empty last workgroup / pass), fused affine + ReLU,
    submanifold and strided (3,1,1) tables."""
    from vision3d_amd.spconv.conv import build_sparse_rulebook, build_subm_rulebook, sparse_conv_forward
    rng = np.random.default_rng(cin + cout + variant)
    coords = kitti_coords(oracle, [5])[:5003]
    shape = [41, 1600, 1408]
    feats = rng.standard_normal((len(coords), cin)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32) * 0.1
    x = make_tensor(coords, feats, shape, 1)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    got = sparse_conv_forward(x.features, dev(w), build_subm_rulebook(x, [3, 3, 3]), dev(sc), dev(sh), True, 4, None, variant,
                              precision).cpu().numpy()
    nbr = oracle.subm_rulebook(coords, shape, 3)
    if precision == "fp32":  # the f16s arithmetic of every variant against float64: fp32-class, strict elementwise bound
        assert_fp32_class(got, conv64(feats, w, nbr, sc, sh, True), f"subm {cin}->{cout} variant {variant} f16s")
    else:
        assert_features_close(got, oracle.sparse_conv_fwd(feats, w, nbr, sc, sh, True), f"subm {cin}->{cout} variant {variant}")
    w1 = (rng.standard_normal((3, 1, 1, cin, cout)) / np.sqrt(cin * 3)).astype(np.float32)
    rb = build_sparse_rulebook(x, [3, 1, 1], [2, 1, 1], [0, 0, 0])
    _, onbr, _ = oracle.sparse_rulebook(coords, shape, [3, 1, 1], [2, 1, 1], [0, 0, 0])
    got = sparse_conv_forward(x.features, dev(w1), rb, None, None, False, 4, None, variant, precision).cpu().numpy()
    if precision == "fp32":
        assert_fp32_class(got, conv64(feats, w1, onbr), f"strided {cin}->{cout} variant {variant} f16s")
    else:
        assert_features_close(got, oracle.sparse_conv_fwd(feats, w1, onbr), f"strided {cin}->{cout} variant {variant}")


@pytest.mark.parametrize("mag", [1e-6, 1.0, 3.0e4])
def test_f16s_scales_follow_the_input_magnitude(oracle, mag):
    """f16s splits x * s with a power-of-two s per tensor: the result must not depend on the tensor's magnitude (features around 1e-6,
    1 and 3e4 -- beyond f16's own range without the scale) and stays fp32-class against float64."""
    from vision3d_amd.spconv.conv import build_subm_rulebook, sparse_conv_forward
    rng = np.random.default_rng(5)
    coords = kitti_coords(oracle, [6])[:3000]
    shape = [41, 1600, 1408]
    feats = (rng.standard_normal((len(coords), 64)) * mag).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 64, 64)) / np.sqrt(64 * 9)).astype(np.float32)
    x = make_tensor(coords, feats, shape, 1)
    got = sparse_conv_forward(x.features, dev(w), build_subm_rulebook(x, [3, 3, 3]), None, None, False, 4, None, 0, "fp32").cpu().numpy()
    assert np.isfinite(got).all()
    assert_fp32_class(got, conv64(feats, w, oracle.subm_rulebook(coords, shape, 3)), f"f16s magnitude {mag}")


def test_tiny_and_empty_inputs(oracle):
    from vision3d_amd import spconv
    conv = spconv.SubMConv3d(4, 16, 3, indice_key="k", bias=False).cuda()
    for n in (0, 1, 63, 65):
        coords = np.stack([np.zeros(n), np.arange(n) % 7, np.arange(n), np.arange(n) * 2], 1).astype(np.int32)
        feats = np.random.default_rng(n).standard_normal((n, 4)).astype(np.float32)
        out = conv(make_tensor(coords, feats, [8, 80, 160], 1))
        assert out.features.shape == (n, 16)
        if n:
            ref = oracle.sparse_conv_fwd(feats, conv.weight.detach().cpu().numpy(), oracle.subm_rulebook(coords, [8, 80, 160], 3))
            assert_features_close(out.features.detach().cpu().numpy(), ref, f"tiny n={n}")
    assert out.dense().shape == (1, 16, 8, 80, 160)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64)])
def test_ring_kernel_edge_sizes(oracle, cin, cout):
    """The default 3x3x3 kernel for 32/64-channel layers (two 16-row tiles per workgroup) at the sizes where its tiling has
    edges: one row, one tile, one workgroup +- 1 row, a dead second tile, an odd number of workgroups; capacity larger than
    the live count (the kernel reads the count from device memory and most workgroups must exit)."""
    from vision3d_amd import spconv
    conv = spconv.SubMConv3d(cin, cout, 3, indice_key="k", bias=False).cuda()
    w = conv.weight.detach().cpu().numpy()
    for n in (1, 15, 16, 17, 31, 32, 33, 48, 95, 161):
        coords = np.stack([np.zeros(n), np.arange(n) % 5, (np.arange(n) // 5) % 40, np.arange(n) // 200 + (np.arange(n) % 3)], 1).astype(np.int32)
        coords = np.unique(coords, axis=0)
        m = len(coords)
        feats = np.random.default_rng(n).standard_normal((m, cin)).astype(np.float32)
        out = conv(make_tensor(coords, feats, [8, 80, 160], 1))
        ref = oracle.sparse_conv_fwd(feats, w, oracle.subm_rulebook(coords, [8, 80, 160], 3))
        assert_features_close(out.features.detach().cpu().numpy(), ref, f"ring n={m}")


def test_ring_kernel_tiles_per_workgroup(oracle):
    """64 -> 64 ring kernel with 2 / 3 / 4 sixteen-row tiles per workgroup (variants 12 / 13 / 14; the plan picks the smallest
    count that keeps a layer inside one round of 256 LDS-filling workgroups): the oracle's result, and the SAME bits from all
    three forms (a row's products are accumulated in the same order whatever tile of whatever workgroup holds it), at sizes
    around every tiling edge."""
    from vision3d_amd.spconv.conv import build_subm_rulebook, sparse_conv_forward
    rng = np.random.default_rng(64)
    all_coords = kitti_coords(oracle, [5])
    shape = [41, 1600, 1408]
    w = (rng.standard_normal((3, 3, 3, 64, 64)) / np.sqrt(64 * 9)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, 64).astype(np.float32), rng.standard_normal(64).astype(np.float32) * 0.1
    for n in (1, 16, 33, 47, 48, 49, 63, 64, 65, 97, 129, 5003):
        coords = all_coords[:n]
        feats = rng.standard_normal((n, 64)).astype(np.float32)
        x = make_tensor(coords, feats, shape, 1)
        rb = build_subm_rulebook(x, [3, 3, 3])
        ref = oracle.sparse_conv_fwd(feats, w, oracle.subm_rulebook(coords, shape, 3), sc, sh, True)
        outs = [sparse_conv_forward(x.features, dev(w), rb, dev(sc), dev(sh), True, 4, None, v).cpu().numpy() for v in (12, 13, 14)]
        assert_features_close(outs[0], ref, f"ring 2 tiles n={n}")
        np.testing.assert_array_equal(outs[0], outs[1], err_msg=f"3 tiles vs 2, n={n}")
        np.testing.assert_array_equal(outs[0], outs[2], err_msg=f"4 tiles vs 2, n={n}")
    # the grid is capped at what the chip holds at once and every workgroup strides over the live tile groups: on 40 003 rows of a
    # 7-frame batch the 2-tile form (1 251 tile groups) takes five trips per workgroup, the 4-tile form three -- same bits, and the
    # 16-row kernel's result within the feature bar
    coords = kitti_coords(oracle, [1, 2, 3, 4, 5, 6, 7])[:40003]
    feats = rng.standard_normal((len(coords), 64)).astype(np.float32)
    x = make_tensor(coords, feats, shape, 7)
    rb = build_subm_rulebook(x, [3, 3, 3])
    big = [sparse_conv_forward(x.features, dev(w), rb, dev(sc), dev(sh), True, 4, None, v).cpu().numpy() for v in (12, 14, 1)]
    np.testing.assert_array_equal(big[0], big[1])
    assert_features_close(big[0], big[2], "ring (strided grid) vs 16-row kernel")


def test_offset_outer_kernel_picks_its_pass_size_from_the_live_row_count(oracle):
    """spconv_fwd_rows_kouter (the 64 -> 64 kernel from 32 k rows): above 65 536 live rows a launch walks 384-row passes (three tiles
    per wave) so that the layer stays inside one round of 256 workgroups, below 256-row passes; the count is device-side.  Same
    bits as the 256-row-pass form (variant 8) on both sides of the switch, and the 16-row kernel's result within the feature bar."""
    from vision3d_amd.spconv.conv import build_subm_rulebook, sparse_conv_forward
    rng = np.random.default_rng(7)
    coords = kitti_coords(oracle, [1, 2, 3, 4, 5, 6, 7])
    assert len(coords) > 70000
    shape = [41, 1600, 1408]
    w = (rng.standard_normal((3, 3, 3, 64, 64)) / np.sqrt(64 * 9)).astype(np.float32)
    for n in (len(coords), 65536 + 17, 65536, 40001):
        c = coords[:n]
        feats = rng.standard_normal((n, 64)).astype(np.float32)
        x = make_tensor(c, feats, shape, 7)
        rb = build_subm_rulebook(x, [3, 3, 3])
        auto = sparse_conv_forward(x.features, dev(w), rb, None, None, True, 4, None, 6).cpu().numpy()
        two = sparse_conv_forward(x.features, dev(w), rb, None, None, True, 4, None, 8).cpu().numpy()
        rows16 = sparse_conv_forward(x.features, dev(w), rb, None, None, True, 4, None, 1).cpu().numpy()
        np.testing.assert_array_equal(auto, two, err_msg=f"n={n}")
        assert_features_close(auto, rows16, f"offset-outer vs 16-row kernel, n={n}")


def test_densify_exact(oracle):
    rng = np.random.default_rng(3)
    coords = kitti_coords(oracle, [5, 6])
    oc, _, osh = oracle.sparse_rulebook(coords, [41, 1600, 1408], 3, 2, 1)
    oc2, _, osh2 = oracle.sparse_rulebook(oc, osh, 3, 2, 1)
    oc3, _, osh3 = oracle.sparse_rulebook(oc2, osh2, 3, 2, [0, 1, 1])
    feats = rng.standard_normal((len(oc3), 64)).astype(np.float32)
    got = make_tensor(oc3, feats, osh3, 2).dense().detach().cpu().numpy()
    np.testing.assert_array_equal(got, oracle.densify(feats, oc3, 2, osh3))


def test_sparse_sequential_fuses_and_matches_unfused(oracle):
    """conv+BN(eval)+ReLU through SparseSequential (one launch) == conv, then torch BN, then ReLU."""
    from gpu_util import randomize_bn
    from vision3d_amd.detector.sparse_cnn import make_sparse_conv_layer, make_subm_layer
    torch.manual_seed(0)
    coords = kitti_coords(oracle, [7])[:5000]
    feats = np.random.default_rng(1).standard_normal((len(coords), 16)).astype(np.float32)
    for layer in (make_subm_layer(16, 32, 3, indice_key="a"), make_sparse_conv_layer(16, 32, 3, 2, padding=1)):
        layer = layer.cuda().eval()
        randomize_bn(layer)
        x = make_tensor(coords, feats, [41, 1600, 1408], 1)
        fused = layer(x)
        raw = layer[0](x)
        with torch.no_grad():
            ref = torch.relu(layer[1](raw.features))
        assert_features_close(fused.features.detach().cpu().numpy(), ref.cpu().numpy(), "fused vs unfused")
        np.testing.assert_array_equal(fused.indices.cpu().numpy(), raw.indices.cpu().numpy())


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("variant", [1, 5, 6, 7, 10, 16],
                         ids=["16rows", "64rows_lds_weights", "offset_outer_staged", "offset_outer_regs", "lds_ring", "lds_ring_regs"])
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 32), (64, 64)])
def test_presplit_rows_every_kernel_variant(oracle, cin, cout, variant, precision):
    """Rows that are ALREADY split into the arithmetic's 16-bit pieces (v3d_sparse_conv_fwd_packed in_split / out_split: what the
    layers of a plan hand each other) through every kernel of the packed product: gathering the split copy gives the bits of
    gathering the fp32 rows, and the split copy a layer writes is the split of the fp32 rows it writes -- ragged tail, fused affine +
    ReLU, f16s scale entries from the tensors' own maxima."""
    from vision3d_amd import _lib as L
    from vision3d_amd.runtime import act_entry_from_tensor, rows_split
    from vision3d_amd.spconv.conv import build_subm_rulebook, pack_sparse_weight
    rng = np.random.default_rng(cin * 3 + cout + variant)
    coords = kitti_coords(oracle, [5])[:5003]
    feats = dev(rng.standard_normal((len(coords), cin)).astype(np.float32))
    w = dev((rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32))
    sc, sh = dev(rng.uniform(0.5, 1.5, cout).astype(np.float32)), dev(rng.standard_normal(cout).astype(np.float32) * 0.1)
    x = make_tensor(coords, feats.cpu().numpy(), [41, 1600, 1408], 1)
    rb = build_subm_rulebook(x, [3, 3, 3])
    f16s = precision == "fp32"
    prec = L.PRECISIONS[precision]
    img = pack_sparse_weight(w, 27, cin, cout, precision)
    entry = act_entry_from_tensor(feats) if f16s else None

    def run(in_rows, in_split, out, next_entry, out_split):
        L.check(L.lib().v3d_sparse_conv_fwd_packed(L.ptr(in_rows), L.ptr(img), L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap, 27, cin, cout,
                                                    L.ptr(sc), L.ptr(sh), 1, L.ptr(out), -variant, prec, L.ptr(entry), L.ptr(next_entry),
                                                    None, L.ptr(in_split), L.ptr(out_split), L.stream_ptr()), "fwd_packed2")
    ref = torch.empty((rb.n, cout), dtype=torch.float32, device="cuda")
    run(feats, None, ref, None, None)
    next_entry = act_entry_from_tensor(ref) if f16s else None
    got = torch.empty_like(ref)
    got_s = torch.zeros((rb.n, 2 * cout), dtype=torch.int16, device="cuda")
    run(None, rows_split(feats, precision, entry), got, next_entry, got_s)     # split rows in, fp32 rows + split rows out
    assert torch.equal(got, ref)
    assert torch.equal(got_s, rows_split(ref, precision, next_entry))
    only_s = torch.zeros_like(got_s)
    run(feats, None, None, next_entry, only_s)                                 # fp32 rows in, split rows ONLY out
    assert torch.equal(only_s, got_s)


def test_scale_entry_grid_form_equals_the_single_workgroup_form():
    """v3d_act_scale_from_rows with a scratch pair (grid of workgroups, self-resetting) against its one-workgroup form (scratch NULL) on the same rows:
    identical entries; a device-side row count, a misaligned view, a one-element tensor; the scratch reads zero again after
    every launch (so back-to-back launches on one stream need no fill)."""
    from vision3d_amd import _lib as L
    torch.manual_seed(3)
    lib = L.lib()
    scratch = L.scale_scratch(torch.device("cuda", torch.cuda.current_device()))
    big = torch.randn(70001, 64, device="cuda") * 3.0
    big[41234, 17] = -913.25  # the maximum sits in one place
    cases = [(big, None, 64), (big[:1], None, 64), (big.reshape(-1)[1:64 * 1000 + 1].reshape(1000, 64), None, 64),
             (big, torch.tensor([12345], dtype=torch.int32, device="cuda"), 64), (torch.zeros(100, 16, device="cuda"), None, 16),
             (torch.full((1, 1), 2.0 ** -130, device="cuda"), None, 1)]
    for ci, (rows, n_dev, c) in enumerate(cases):
        for headroom in (0, 5):
            a = torch.empty(4, device="cuda")
            b = torch.empty(4, device="cuda")
            L.check(lib.v3d_act_scale_from_rows(L.ptr(rows), L.ptr(n_dev), rows.shape[0], c, headroom, L.ptr(a), None, L.stream_ptr()), "one")
            for _ in range(2):  # twice: the second launch runs on the scratch the first left behind
                b.fill_(-1)
                L.check(lib.v3d_act_scale_from_rows(L.ptr(rows), L.ptr(n_dev), rows.shape[0], c, headroom, L.ptr(b), L.ptr(scratch),
                                                     L.stream_ptr()), "grid")
                assert torch.equal(a, b), (ci, headroom, a.tolist(), b.tolist())
                assert scratch.tolist() == [0, 0]
            n = rows.shape[0] if n_dev is None else int(n_dev.item())
            amax = float(rows[:n].abs().max())
            assert float(a[3]) == amax
            if amax > 1e-30:
                assert 2.0 ** (13 - headroom) <= amax * float(a[0]) < 2.0 ** (14 - headroom), (ci, amax, a.tolist())
