"""GPU parity: the split-precision MFMA convolution (csrc/dense_conv.hip) vs fp32 / float64 torch convolutions -- standalone, as
the whole RPN + heads stack, and end to end.  Both arithmetics: "fp32" (f16s, the default of the inference paths: fp32-class,
strict elementwise bound against float64) and "bf16x3" (features within the repository's 1e-4 bar)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import FP32_CLASS_FLOOR, STRICT_FP32_CLASS, assert_features_close, assert_fp32_class, randomize_bn, strict_rel_err
from vision3d_amd import synth
from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


def _planes_to_float(y_hi, y_lo, precision, entry=None):
    if precision == "bf16x3":
        return y_hi.view(torch.bfloat16).float() + y_lo.view(torch.bfloat16).float()
    return (y_hi.view(torch.float16).float() + y_lo.view(torch.float16).float()) * entry[1]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("b,h,w,cin,cout,k", [(2, 37, 29, 64, 128, 3), (1, 20, 16, 128, 128, 3), (1, 33, 50, 128, 128, 1),
                                             (3, 9, 7, 32, 256, 3), (1, 40, 31, 128, 16, 1), (1, 61, 53, 256, 128, 1),
                                             (2, 24, 12, 128, 256, 3), (1, 30, 44, 192, 128, 3), (1, 19, 23, 64, 64, 3), (1, 25, 17, 96, 128, 1)])
def test_conv_matches_torch_fp32(b, h, w, cin, cout, k, precision):
    """The shape list reaches all three kernels through the dispatch of v3d_conv2d_nhwc_split: the 144-pixel kernel (Cin >= 64,
    Cout > 32, W >= 16), the 64-pixel kernel (Cin = 32, Cout <= 32, or W < 16: the (2, 24, 12, 128, 256) and (3, 9, 7, 32, 256)
    rows) and the streaming 1x1 head kernel (128 -> 16).  f16s: the input's scale entry from the tensor's own maximum, the
    output planes' from the reference's; the result must be fp32-class against float64 (strict elementwise bound)."""
    from vision3d_amd.runtime import conv2d_split, pack_conv_weight, scale_entry_from_max, to_split_nhwc
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    x = torch.randn(b, cin, h, w, generator=g).cuda()
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
    scale = (torch.rand(cout, generator=g) + 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda() * 0.2
    ref = F.relu(F.conv2d(x.double(), (wt * scale.view(-1, 1, 1, 1)).double(), bias.double(), padding=k // 2))
    f16s = precision == "fp32"
    hi, lo = to_split_nhwc(x, precision)
    out_entry = scale_entry_from_max(ref.abs().max().float(), 1) if f16s else None
    pr = (hi.v3d_entry, out_entry, None) if f16s else None
    img = pack_conv_weight(wt, scale, precision)
    (y_hi, y_lo), y = conv2d_split(hi, lo, img, bias, True, cin, cout, k, out_split=(cout % 8 == 0), out_nchw=True, pr=pr)

    # the yardstick of "fp32-class": torch's own fp32 convolution of the same operands against float64 (random-signed inputs cancel
    # harder than post-ReLU activations: fp32 summation noise alone reaches 1-2e-4 on entries 1000 x below the maximum)
    torch.backends.cudnn.allow_tf32 = False
    ref32 = F.relu(F.conv2d(x, wt * scale.view(-1, 1, 1, 1), bias, padding=k // 2))
    torch32 = strict_rel_err(ref32.cpu().numpy(), ref.cpu().numpy())

    def check(got, want, what, bound=max(STRICT_FP32_CLASS, 2.0 * torch32)):
        if f16s:
            assert_features_close(got, want, what, floor=FP32_CLASS_FLOOR)
            assert strict_rel_err(got, want) < bound, (what, strict_rel_err(got, want), "torch fp32:", torch32)
        else:
            assert_features_close(got, want, what)
    check(y.cpu().numpy(), ref.cpu().numpy(), f"conv {cin}->{cout} k{k} fp32 NCHW output")
    if y_hi is not None:  # split planes: hi + lo reproduces the fp32 value (bf16: to ~2^-17, f16s: to ~2^-22)
        back = _planes_to_float(y_hi, y_lo, precision, out_entry).permute(0, 3, 1, 2)
        check(back.cpu().numpy(), ref.cpu().numpy(), "split planes output")
    # without bias / relu
    pr2 = (hi.v3d_entry, None, None) if f16s else None
    _, y2 = conv2d_split(hi, lo, pack_conv_weight(wt, None, precision), None, False, cin, cout, k, out_split=False, out_nchw=True, pr=pr2)
    plain64 = F.conv2d(x.double(), wt.double(), None, padding=k // 2).cpu().numpy()
    plain32 = strict_rel_err(F.conv2d(x, wt, None, padding=k // 2).cpu().numpy(), plain64)
    check(y2.cpu().numpy(), plain64, "plain conv", max(STRICT_FP32_CLASS, 2.0 * plain32))


@pytest.mark.parametrize("mag", [1e-5, 1.0, 1e4])
def test_f16s_conv_is_magnitude_invariant_and_flags_out_of_range_outputs(mag):
    """Power-of-two scales are exact: inputs of size 1e-5 / 1 / 1e4 give the fp32-class result.  An output entry calibrated for
    a 1000 x smaller tensor (limit = 2^15 / s exceeded) raises the range flag to 2; inside the limit the flag stays untouched."""
    from vision3d_amd.runtime import conv2d_split, pack_conv_weight, scale_entry_from_max, to_split_nhwc
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1, 128, 20, 32, generator=g) * mag).cuda()
    wt = (torch.randn(128, 128, 3, 3, generator=g) / (128 * 9) ** 0.5).cuda()
    ref = F.relu(F.conv2d(x.double(), wt.double(), None, padding=1))
    hi, lo = to_split_nhwc(x, "fp32")
    img = pack_conv_weight(wt, None, "fp32")
    flag = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    entry = scale_entry_from_max(ref.abs().max().float(), 2)
    (y_hi, y_lo), _ = conv2d_split(hi, lo, img, None, True, 128, 128, 3, pr=(hi.v3d_entry, entry, flag))
    got = _planes_to_float(y_hi, y_lo, "fp32", entry).permute(0, 3, 1, 2).cpu().numpy()
    assert int(flag) == -1
    assert_features_close(got, ref.cpu().numpy(), f"f16s magnitude {mag}", floor=FP32_CLASS_FLOOR)
    assert strict_rel_err(got, ref.cpu().numpy()) < STRICT_FP32_CLASS
    small = scale_entry_from_max(ref.abs().max().float() / 1000.0, 0)
    conv2d_split(hi, lo, img, None, True, 128, 128, 3, pr=(hi.v3d_entry, small, flag))
    assert int(flag) == 2


def build_model(seed=0):
    from vision3d_amd.detector import Second
    torch.manual_seed(seed)
    model = Second(second_car_cfg())
    randomize_bn(model, seed)
    with torch.no_grad():
        model.head.conv_cls.weight.normal_(0, 0.05)
        model.head.conv_reg.weight.normal_(0, 0.02)
    return model.cuda().eval()


def test_dense_head_stack_matches_torch():
    """Real BEV map -> 7 RPN convs + heads: MFMA path vs torch fp32 modules."""
    from vision3d_amd.runtime import to_split_nhwc
    model = build_model(1)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (0, 1)]
    with torch.no_grad():
        bev = model.bev_from_points(clouds)
        ref_feat = model.rpn.up_block(model.rpn.down_block(bev))
        ref_cls, ref_reg = model.head(ref_feat)
        maps, feats = model.dense_plan().forward(*to_split_nhwc(bev, model.precision), want_features=True)
        cls_map, reg_map = model.head.maps_from_fused(maps)
        cls2, reg2 = model.head_maps_from_points(clouds)   # + the split densify path
    assert_features_close(feats.cpu().numpy(), ref_feat.cpu().numpy(), "RPN features")
    assert_features_close(cls_map.cpu().numpy(), ref_cls.cpu().numpy(), "cls map")
    assert_features_close(reg_map.cpu().numpy(), ref_reg.cpu().numpy(), "reg map")
    # the split densify path: bit-identical in bf16x3 (the same pieces either way); in f16s the two paths split the BEV map under
    # different scale entries (the tensor's own maximum vs the plan's calibrated entry with headroom), which moves the lo pieces
    # of small values by the 2^-24 subnormal quantum: equal to fp32 summation noise, not to the bit
    if model.precision == "bf16x3":
        np.testing.assert_array_equal(cls2.cpu().numpy(), cls_map.cpu().numpy())
        np.testing.assert_array_equal(reg2.cpu().numpy(), reg_map.cpu().numpy())
    else:
        assert_features_close(cls2.cpu().numpy(), cls_map.cpu().numpy(), "cls map, split densify", floor=FP32_CLASS_FLOOR)
        assert_features_close(reg2.cpu().numpy(), reg_map.cpu().numpy(), "reg map, split densify", floor=FP32_CLASS_FLOOR)


def test_native_path_matches_cpu_oracle_end_to_end():
    from gpu_util import numpy_state_dict
    from oracle import second_cpu
    cfg = second_car_cfg()
    model = build_model(2)
    cloud = synth.make_cloud(3)
    with torch.no_grad():
        cls_map, reg_map = model.head_maps_from_points([torch.from_numpy(cloud).cuda()])
    ref = second_cpu.second_forward(numpy_state_dict(model), [cloud], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY,
                                    cfg.MAX_VOXELS)
    ref64 = second_cpu.second_forward64(numpy_state_dict(model), [cloud], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS)
    assert_fp32_class(cls_map.cpu().numpy().reshape(ref["cls"].shape), ref["cls"], "P_cls native vs oracle", ref64["cls"])
    reg = reg_map.permute(0, 1, 5, 2, 3, 4).reshape(ref["reg"].shape)
    assert_fp32_class(reg.cpu().numpy(), ref["reg"], "P_reg native vs oracle", ref64["reg"])


def test_inference_points_paths_agree():
    from vision3d_amd.core import AnchorGenerator
    model = build_model(3)
    anchors = AnchorGenerator(second_car_cfg()).anchors.cuda()
    clouds = [torch.from_numpy(synth.make_cloud(5)).cuda()]
    with torch.no_grad():
        a = model.inference_points(clouds, anchors)                     # device proposal stage
        b = model.inference_points(clouds, anchors, proposals="torch")  # op-by-op torch statement of the proposal stage
        with pytest.raises(ValueError):
            model.inference_points(clouds, anchors, dense="torch")      # the MIOpen dense path no longer exists
    assert a[0].shape[1] == 7 and abs(len(a[0]) - len(b[0])) <= max(2, len(b[0]) // 10)


def _occupancy_numpy(occ, b, h, w):
    """inverted bitmap (B * H, ceil(W / 32)) int32 -> bool (B, H, W) occupied"""
    words = occ.cpu().numpy().astype(np.uint32)
    bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(words.shape[0], -1)[:, :w]
    return (bits == 0).reshape(b, h, w)


def _tiles_with_background(occupied, reach, tile=80):
    """fraction of `tile`-pixel runs (flattened B*H*W order) whose every pixel is further than `reach` from all occupied ones"""
    b, h, w = occupied.shape
    near = np.zeros_like(occupied)
    padded = np.pad(occupied, ((0, 0), (reach, reach), (reach, reach)))
    for dy in range(2 * reach + 1):
        for dx in range(2 * reach + 1):
            near |= padded[:, dy:dy + h, dx:dx + w]
    flat = near.reshape(-1)
    flat = np.concatenate([flat, np.zeros((-len(flat)) % tile, dtype=bool)])
    return float((~flat.reshape(-1, tile).any(1)).mean())


def test_bev_occupancy_bitmap_matches_numpy():
    """v3d_bev_occupancy_bits and the plan's own bitmap (written by its densify kernel): bit (b, y, x) cleared <=> occupied."""
    from vision3d_amd import _lib as L
    rng = np.random.default_rng(0)
    B, H, W, n, cap = 2, 23, 75, 60, 64
    coords = np.stack([rng.integers(0, B, n), rng.integers(0, 2, n), rng.integers(0, H, n), rng.integers(0, W, n)], 1).astype(np.int32)
    coords[0] = (0, 0, 0, 0)
    coords[1] = (1, 1, H - 1, W - 1)
    c = torch.zeros((cap, 4), dtype=torch.int32, device="cuda")
    c[:n] = torch.from_numpy(coords).cuda()
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    occ = torch.empty((B * H, (W + 31) // 32), dtype=torch.int32, device="cuda")
    assert L.lib().v3d_bev_occupancy_words(B, H, W) == occ.numel()
    L.check(L.lib().v3d_bev_occupancy_bits(L.ptr(c), L.ptr(n_dev), cap, B, H, W, L.ptr(occ), L.stream_ptr()), "bev_occupancy_bits")
    ref = np.zeros((B, H, W), dtype=bool)
    ref[coords[:, 0], coords[:, 2], coords[:, 3]] = True
    np.testing.assert_array_equal(_occupancy_numpy(occ, B, H, W), ref)
    model = build_model(3)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (5, 6)]
    with torch.no_grad():
        plan, flat, offsets = model._plan_for(clouds)
        plan.forward_split(flat, offsets)
        _, sites, n_rows, shape = plan.layer_output(len(plan.layers) - 1)
        sites = sites[:int(n_rows)].cpu().numpy()
        ref = np.zeros((2, shape[1], shape[2]), dtype=bool)
        ref[sites[:, 0], sites[:, 2], sites[:, 3]] = True
        np.testing.assert_array_equal(_occupancy_numpy(plan.bev_occupancy(2), 2, shape[1], shape[2]), ref)


def test_background_left_in_place_across_frames_is_bit_identical():
    """DenseHeadState: persistent RPN planes + tile states.  A sequence of different frames (and a repeated one) through the same
    state gives, frame by frame, exactly the maps of the all-tiles forward; tiles that were background in the frame before are
    not written (their state word stays 0), tiles that turn live / background flip their word."""
    model = build_model(3)
    dense = model.dense_plan()
    state = dense.new_state(torch.device("cuda"))
    seeds = [0, 1, 1, 5, 0]
    with torch.no_grad():
        prev_live = None
        for j, seed in enumerate(seeds):
            clouds = [torch.from_numpy(synth.make_cloud(seed)).cuda()]
            plan, flat, offsets = model._plan_for(clouds)
            hi, lo = plan.forward_split(flat, offsets)
            occ = plan.bev_occupancy(1).clone()
            full = dense.forward(hi, lo)
            got = dense.forward(hi, lo, occ=occ, work=state)
            assert torch.equal(full, got), f"frame {j} (seed {seed})"
            # (the layers of a persistent state zero each other's tile counters at their start: the pairs are not zero between frames)
            # (with the fused tail -- the default -- the 1x1 up-conv runs inside the head's launch on every pixel: its planes and
            # tile states are not used)
            live = [t.clone() for t in (state.tiles[:-1] if dense.fuse_tail else state.tiles)]
            for t in live:
                assert set(t.unique().tolist()) <= {0, 1}
                assert 0 < int(t.sum()) < t.numel()  # a sparse map: some tiles convolve, some hold the response
            if j > 0 and seeds[j] == seeds[j - 1]:
                assert all(torch.equal(a, b) for a, b in zip(live, prev_live))  # same frame again: same set of live tiles
            prev_live = live
        # the state follows the weights: a changed parameter re-creates the planes with every tile marked "not in place"
        next(m for m in model.rpn.down_block if isinstance(m, torch.nn.Conv2d)).weight.mul_(1.01)  # (under no_grad: bumps the version)
        clouds = [torch.from_numpy(synth.make_cloud(2)).cuda()]
        plan, flat, offsets = model._plan_for(clouds)
        hi, lo = plan.forward_split(flat, offsets)
        occ = plan.bev_occupancy(1).clone()
        assert torch.equal(dense.forward(hi, lo), dense.forward(hi, lo, occ=occ, work=state))


@pytest.mark.parametrize("frames", [(0,), (3, 4)])
def test_background_skipping_is_bit_identical_and_skips(frames):
    """RPN with tiles far from every occupied pixel copied from the empty-map response == RPN with every tile convolved."""
    from vision3d_amd.runtime import conv2d_split
    model = build_model(3)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in frames]
    with torch.no_grad():
        plan, flat, offsets = model._plan_for(clouds)
        hi, lo = plan.forward_split(flat, offsets)
        occ = plan.bev_occupancy(len(clouds)).clone()
        dense = model.dense_plan()
        full = dense.forward(hi, lo)
        skipped = dense.forward(hi, lo, occ=occ)
        assert torch.equal(full, skipped)
        # layer by layer (split planes), and how much the sparse map lets the kernel skip
        occupied = _occupancy_numpy(occ, len(clouds), hi.shape[1], hi.shape[2])
        bg = dense.background(hi.shape[1], hi.shape[2], hi.device)
        x_hi, x_lo, reach = hi, lo, 0
        for i, ly in enumerate(dense.layers[:-1]):
            reach += ly["k"] // 2
            pr = dense._pr(i, hi.v3d_entry, None)  # (f16s: the layer's scale entries; None for bf16x3)
            (a_hi, a_lo), _ = conv2d_split(x_hi, x_lo, ly["img"], ly["bias"], ly["relu"], ly["cin"], ly["cout"], ly["k"], pr=pr)
            (b_hi, b_lo), _ = conv2d_split(x_hi, x_lo, ly["img"], ly["bias"], ly["relu"], ly["cin"], ly["cout"], ly["k"],
                                           occ=occ, reach=reach, bg=bg[i], pr=pr)
            assert torch.equal(a_hi, b_hi) and torch.equal(a_lo, b_lo), f"layer {i}"
            work = torch.zeros(2, dtype=torch.int32, device="cuda")  # persistent grid drawing 80-pixel tiles from a counter
            (c_hi, c_lo), _ = conv2d_split(x_hi, x_lo, ly["img"], ly["bias"], ly["relu"], ly["cin"], ly["cout"], ly["k"],
                                           occ=occ, reach=reach, bg=bg[i], work=work, pr=pr)
            assert torch.equal(a_hi, c_hi) and torch.equal(a_lo, c_lo), f"layer {i} (persistent)"
            assert int(work.abs().sum()) == 0, "the counter pair resets itself"
            assert _tiles_with_background(occupied, reach) > 0.2, f"layer {i}: a sparse BEV map should leave background tiles"
            x_hi, x_lo = a_hi, a_lo
    model.skip_background = False
    with torch.no_grad():
        maps_off = model.fused_head_from_points(clouds)
    model.skip_background = True
    with torch.no_grad():
        maps_on = model.fused_head_from_points(clouds)
    assert torch.equal(maps_on, maps_off)


@pytest.mark.parametrize("b,h,w", [(1, 200, 176), (3, 21, 40), (2, 16, 16), (1, 9, 300)])
def test_background_skipping_random_occupancy_and_shapes(b, h, w):
    """Tile / row / image-boundary geometry of the skip test: random sparse maps at sizes where a 144-pixel tile spans
    several rows or images; a few isolated occupied pixels, some on the borders."""
    from vision3d_amd.runtime import conv2d_split, pack_conv_weight, to_split_nhwc
    g = torch.Generator().manual_seed(b * 1000 + h + w)
    cin = cout = 128
    wt1 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda()
    wt2 = (torch.randn(cout, cout, 3, 3, generator=g) / (cout * 9) ** 0.5).cuda()
    bias1, bias2 = torch.randn(cout, generator=g).cuda() * 0.2, torch.randn(cout, generator=g).cuda() * 0.2
    img1, img2 = pack_conv_weight(wt1), pack_conv_weight(wt2)
    x = torch.zeros(b, cin, h, w)
    n_sites = max(2, b * h * w // 400)
    ys, xs, bs = torch.randint(0, h, (n_sites,), generator=g), torch.randint(0, w, (n_sites,), generator=g), torch.randint(0, b, (n_sites,), generator=g)
    ys[0], xs[0], bs[0] = 0, 0, 0
    ys[1], xs[1], bs[1] = h - 1, w - 1, b - 1
    x[bs, :, ys, xs] = torch.randn(n_sites, cin, generator=g)
    x = x.cuda()
    occ = torch.full((b * h, (w + 31) // 32), -1, dtype=torch.int32)
    occ_np = occ.numpy().view(np.uint32)
    for bb, yy, xx in zip(bs.tolist(), ys.tolist(), xs.tolist()):
        occ_np[bb * h + yy, xx // 32] &= ~np.uint32(1 << (xx % 32))
    occ = occ.cuda()
    hi, lo = to_split_nhwc(x)
    z_hi, z_lo = to_split_nhwc(torch.zeros(1, cin, h, w, device="cuda"))
    (bg1_hi, bg1_lo), _ = conv2d_split(z_hi, z_lo, img1, bias1, True, cin, cout, 3)
    (bg2_hi, bg2_lo), _ = conv2d_split(bg1_hi, bg1_lo, img2, bias2, True, cout, cout, 3)
    (a1_hi, a1_lo), _ = conv2d_split(hi, lo, img1, bias1, True, cin, cout, 3)
    (a2_hi, a2_lo), _ = conv2d_split(a1_hi, a1_lo, img2, bias2, True, cout, cout, 3)
    (s1_hi, s1_lo), _ = conv2d_split(hi, lo, img1, bias1, True, cin, cout, 3, occ=occ, reach=1, bg=(bg1_hi, bg1_lo))
    (s2_hi, s2_lo), _ = conv2d_split(s1_hi, s1_lo, img2, bias2, True, cout, cout, 3, occ=occ, reach=2, bg=(bg2_hi, bg2_lo))
    assert torch.equal(a1_hi, s1_hi) and torch.equal(a1_lo, s1_lo)
    assert torch.equal(a2_hi, s2_hi) and torch.equal(a2_lo, s2_lo)
    work = torch.zeros(2, dtype=torch.int32, device="cuda")
    (p1_hi, p1_lo), _ = conv2d_split(hi, lo, img1, bias1, True, cin, cout, 3, occ=occ, reach=1, bg=(bg1_hi, bg1_lo), work=work)
    (p2_hi, p2_lo), _ = conv2d_split(p1_hi, p1_lo, img2, bias2, True, cout, cout, 3, occ=occ, reach=2, bg=(bg2_hi, bg2_lo), work=work)
    assert torch.equal(a1_hi, p1_hi) and torch.equal(a1_lo, p1_lo)
    assert torch.equal(a2_hi, p2_hi) and torch.equal(a2_lo, p2_lo)
    assert int(work.abs().sum()) == 0


def test_background_skipping_on_an_empty_and_a_full_map():
    """No occupied pixel: every tile is background (the result is the empty-map response itself).  Every pixel occupied:
    nothing is skipped."""
    from vision3d_amd.runtime import to_split_nhwc
    model = build_model(4)
    dense = model.dense_plan()
    h, w = 200, 176
    with torch.no_grad():
        hi, lo = to_split_nhwc(torch.zeros(1, 128, h, w, device="cuda"), model.precision)
        none = torch.full((h, (w + 31) // 32), -1, dtype=torch.int32, device="cuda")
        assert torch.equal(dense.forward(hi, lo), dense.forward(hi, lo, occ=none))
        x = torch.randn(1, 128, h, w, device="cuda")
        hi, lo = to_split_nhwc(x, model.precision)
        dense.recalibrate()  # (f16s: the entries of the all-zero map above do not fit this one)
        every = torch.zeros((h, (w + 31) // 32), dtype=torch.int32, device="cuda")
        assert torch.equal(dense.forward(hi, lo), dense.forward(hi, lo, occ=every))


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("b,h,w,cout2", [(1, 200, 176, 16), (2, 37, 29, 14), (1, 5, 3, 16), (3, 64, 40, 8)])
def test_fused_1x1_and_head_equal_the_two_launches_bit_for_bit(b, h, w, cout2, precision):
    """v3d_conv2d_1x1_head_fused (the RPN's 1x1 up-conv + ReLU and the fused [cls | reg] head on top, one pass over the pixels) against
    the two launches it replaces -- the tile kernel writing split planes, then the streaming head kernel: identical bits, at the
    KITTI map size, at sizes with a ragged last 16-pixel tile, with fewer than 16 head channels and at batch > 1."""
    from vision3d_amd import _lib as L
    from vision3d_amd.runtime import _prec_struct, conv2d_split, pack_conv_weight, scale_entry_from_max, to_split_nhwc
    g = torch.Generator().manual_seed(b * 1000 + h + cout2)
    x = torch.randn(b, 128, h, w, generator=g).cuda()
    w1 = (torch.randn(128, 128, 1, 1, generator=g) / 128 ** 0.5).cuda()
    s1 = (torch.rand(128, generator=g) + 0.5).cuda()
    b1 = (torch.randn(128, generator=g) * 0.2).cuda()
    w2 = (torch.randn(cout2, 128, 1, 1, generator=g) / 128 ** 0.5).cuda()
    b2 = (torch.randn(cout2, generator=g) * 0.2).cuda()
    f16s = precision == "fp32"
    mid64 = F.relu(F.conv2d(x.double(), (w1 * s1.view(-1, 1, 1, 1)).double(), b1.double()))
    hi, lo = to_split_nhwc(x, precision)
    mid_entry = scale_entry_from_max(mid64.abs().max().float(), 1) if f16s else None
    img1, img2 = pack_conv_weight(w1, s1, precision), pack_conv_weight(w2, None, precision)
    (m_hi, m_lo), _ = conv2d_split(hi, lo, img1, b1, True, 128, 128, 1, out_split=True, out_nchw=False,
                                   pr=(hi.v3d_entry, mid_entry, None) if f16s else None)
    _, two = conv2d_split(m_hi, m_lo, img2, b2, False, 128, cout2, 1, out_split=False, out_nchw=True,
                          pr=(mid_entry, None, None) if f16s else None)
    fused = torch.empty_like(two)
    ref_struct, keep = _prec_struct((hi.v3d_entry, mid_entry, None) if f16s else None)
    L.check(L.lib().v3d_conv2d_1x1_head_fused(L.ptr(hi), L.ptr(lo), L.ptr(img1), L.ptr(b1), 1, L.ptr(img2), L.ptr(b2), 0, b, h, w, 128,
                                               cout2, L.ptr(fused), ref_struct, L.stream_ptr()), "fused")
    torch.cuda.synchronize()
    assert torch.equal(fused, two)
    ref = F.conv2d(mid64, w2.double(), b2.double())
    assert_features_close(fused.cpu().numpy(), ref.cpu().numpy(), "fused tail vs float64", floor=FP32_CLASS_FLOOR if f16s else 1e-4)
    if f16s:
        assert strict_rel_err(fused.cpu().numpy(), ref.cpu().numpy()) < STRICT_FP32_CLASS


def test_dense_head_plan_with_and_without_the_fused_tail():
    """DenseHeadPlan.forward with the fused tail (the default) and with the two launches: same head maps, bit for bit, with and
    without background skipping on a real frame's BEV map."""
    from vision3d_amd.detector import Second
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = Second(cfg).cuda().eval()
    randomize_bn(model, 1)
    clouds = [torch.from_numpy(synth.make_cloud(3)).cuda()]
    with torch.no_grad():
        plan, flat, offsets = model._plan_for(clouds)
        hi, lo = plan.forward_split(flat, offsets)
        dense = model.dense_plan()
        occ = plan.bev_occupancy(1)
        outs = {}
        for fuse in (True, False):
            dense.fuse_tail = fuse
            outs[fuse] = (dense.forward(hi, lo).clone(), dense.forward(hi, lo, occ=occ).clone())
        dense.fuse_tail = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    assert torch.equal(outs[True][0], outs[True][1])


@pytest.mark.parametrize("shape", [(2, 128, 13, 37), (1, 6, 5, 7), (3, 192, 9, 70), (1, 5, 4, 6)])
def test_nchw_split_conversions_are_exact_and_round_trip(shape):
    """v3d_nchw_to_split_nhwc (tiled form for even C, plain form for odd C) against the definition -- hi = RNE(x) as bf16, lo = RNE(x - hi),
    per pixel and channel -- and v3d_split_nhwc_to_nchw back: hi + lo in fp32, exactly; f16s planes hold the pieces of x * s."""
    from vision3d_amd import _lib as L
    from vision3d_amd.runtime import to_split_nhwc
    b, c, h, w = shape
    torch.manual_seed(9)
    x = torch.randn(shape, device="cuda") * 5.0
    x[0, 0, 0, 0] = 0.0
    hi, lo = to_split_nhwc(x)
    ref_hi = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    ref_lo = (x.permute(0, 2, 3, 1) - ref_hi.float()).to(torch.bfloat16)
    assert torch.equal(hi.view(torch.bfloat16), ref_hi) and torch.equal(lo.view(torch.bfloat16), ref_lo)
    if c % 2 == 0:
        back = torch.full(shape, float("nan"), device="cuda")
        L.check(L.lib().v3d_split_nhwc_to_nchw(L.ptr(hi), L.ptr(lo), b, c, h, w, L.ptr(back), L.stream_ptr()), "split_nhwc_to_nchw")
        assert torch.equal(back, (ref_hi.float() + ref_lo.float()).permute(0, 3, 1, 2))
        assert float((back - x).abs().max()) <= 2.0 ** -16 * float(x.abs().max())
    hi16, lo16 = to_split_nhwc(x, "fp32")
    s = float(hi16.v3d_entry[0])
    r_hi = (x.permute(0, 2, 3, 1) * s).to(torch.float16)
    r_lo = (x.permute(0, 2, 3, 1) * s - r_hi.float()).to(torch.float16)
    assert torch.equal(hi16.view(torch.float16), r_hi) and torch.equal(lo16.view(torch.float16), r_lo)
