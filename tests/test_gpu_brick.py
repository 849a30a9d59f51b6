"""GPU parity of csrc/brick.hip (round 6: submanifold convolution over spatially ordered rows, the pass's neighbourhood in LDS):
the planner's tables against numpy, the convolution bit for bit against the offset-outer kernel (same products, same order) and
against float64, for rows in a spatial order (every pass runs the LDS body) and in the first-touch order of a shuffled sweep
(every pass exceeds the LDS slots and runs the direct-gather body)."""
import ctypes

import numpy as np
import pytest
import torch

from gpu_util import FP32_CLASS_FLOOR, STRICT_FP32_CLASS, assert_features_close, dev, strict_rel_err
from vision3d_amd import synth

pytestmark = pytest.mark.gpu
K = 27
UMAX = 480


def _stage2_sites(oracle, order):
    """active sites of SECOND's third stage for a KITTI sweep: ~8 k rows, ~10 neighbours each"""
    from oracle import second_cpu
    cloud = synth.make_cloud(3, order=order)
    _, coords, _ = second_cpu.voxelize_batch([cloud], [0.05, 0.05, 0.1], synth.KITTI_BOUNDS, 5, 20000)
    shape = [41, 1600, 1408]
    for ks, st, pd in (([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1])):
        coords, _, shape = oracle.sparse_rulebook(coords, shape, ks, st, pd)
    return coords, shape


def _tables(L, nbr_dev, n_dev, cap):
    sz = [ctypes.c_size_t() for _ in range(4)]
    L.lib().v3d_sparse_brick_table_bytes(cap, K, *[ctypes.byref(s) for s in sz])
    tabs = [torch.zeros(int(s.value), dtype=torch.uint8, device="cuda") for s in sz]
    L.check(L.lib().v3d_sparse_brick_plan(L.ptr(nbr_dev), L.ptr(n_dev), cap, K, *[L.ptr(t) for t in tabs], L.stream_ptr()), "brick_plan")
    return tabs


@pytest.mark.parametrize("order", ["morton", "shuffled"])
def test_planner_tables_match_numpy(oracle, order):
    from vision3d_amd import _lib as L
    coords, shape = _stage2_sites(oracle, order)
    nbr = oracle.subm_rulebook(coords, shape, 3)  # (n, 27)
    n = nbr.shape[0]
    nbr_dev = dev(np.ascontiguousarray(nbr.T), torch.int32)
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    lidx, ulist, ucnt, tmask = _tables(L, nbr_dev, n_dev, n)
    torch.cuda.synchronize()
    stride = (n + 63) & ~63
    lidx = lidx.view(torch.int16).cpu().numpy().view(np.uint16)[: K * stride].reshape(K, stride)
    npass = (n + 255) // 256
    ucnt = ucnt.view(torch.int32).cpu().numpy()[:npass]
    ulist = ulist.view(torch.int32).cpu().numpy()[: npass * UMAX].reshape(npass, UMAX)
    tmask = tmask.view(torch.int32).cpu().numpy().view(np.uint32)[: (n + 15) // 16]
    rows = np.arange(n)
    pos = (rows & ~63) + 4 * (rows & 15) + ((rows >> 4) & 3)  # a wave's four tiles interleaved
    for p in range(npass):
        blk = nbr[p * 256:(p + 1) * 256]
        u = np.unique(blk[blk >= 0])
        assert ucnt[p] == len(u)
        np.testing.assert_array_equal(ulist[p, :min(len(u), UMAX)], u[:UMAX])
        if len(u) <= 0xFFFE:
            want = np.where(blk >= 0, np.searchsorted(u, np.maximum(blk, 0)), 0xFFFF).astype(np.uint16)  # (rows, 27)
            got = lidx[:, pos[p * 256:p * 256 + len(blk)]].T
            np.testing.assert_array_equal(got, want)
    pad = (-n) % 16
    m = np.concatenate([nbr >= 0, np.zeros((pad, K), bool)]) if pad else nbr >= 0
    want_mask = (m.reshape(-1, 16, K).any(1) * (1 << np.arange(K, dtype=np.uint64))).sum(1).astype(np.uint32)
    np.testing.assert_array_equal(tmask, want_mask)
    if order == "morton":
        assert ucnt.max() <= UMAX  # a spatial order keeps every neighbourhood inside the LDS slots
    else:
        assert (ucnt > UMAX).mean() > 0.5  # first-touch order of a shuffled sweep: the direct-gather body


@pytest.mark.parametrize("order", ["morton", "shuffled"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("cin", [64, 32])
def test_brick_conv_same_bits_as_the_offset_outer_kernel(oracle, order, precision, cin):
    from test_gpu_sparse_conv import conv64
    from vision3d_amd import _lib as L
    from vision3d_amd.spconv.conv import pack_sparse_weight
    lib = L.lib()
    coords, shape = _stage2_sites(oracle, order)
    nbr = oracle.subm_rulebook(coords, shape, 3)
    n = nbr.shape[0]
    rng = np.random.default_rng(cin)
    feats = (rng.standard_normal((n, cin)) * (rng.random((n, cin)) > 0.4)).astype(np.float32)
    w = (rng.standard_normal((K, cin, cin)) / np.sqrt(K * cin / 2)).astype(np.float32)
    scale = (rng.random(cin) + 0.5).astype(np.float32)
    shift = (rng.standard_normal(cin) * 0.1).astype(np.float32)
    prec = L.PRECISIONS[precision]
    nbr_dev = dev(np.ascontiguousarray(nbr.T), torch.int32)
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    f_dev, sc, sh = dev(feats), dev(scale), dev(shift)
    img = pack_sparse_weight(dev(w), K, cin, cin, precision)
    ent_in = torch.empty(4, device="cuda")
    ent_next = torch.tensor([1.0, 1.0, 32768.0, 0.0], device="cuda")
    flag = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    L.check(lib.v3d_act_scale_from_rows(L.ptr(f_dev), None, n, cin, 0, L.ptr(ent_in), None, L.stream_ptr()), "scale")
    fsplit = torch.empty((n, 2 * cin), dtype=torch.int16, device="cuda")
    L.check(lib.v3d_sparse_rows_split(L.ptr(f_dev), L.ptr(n_dev), n, cin, prec, L.ptr(ent_in), L.ptr(fsplit), L.stream_ptr()), "split")
    tabs = _tables(L, nbr_dev, n_dev, n)
    out_a, out_b = torch.zeros((n, cin), device="cuda"), torch.zeros((n, cin), device="cuda")
    os_a = torch.zeros((n, 2 * cin), dtype=torch.int16, device="cuda")
    os_b = torch.zeros_like(os_a)
    variant = 6 if cin == 64 else 5  # offset-outer / 64-row kernel: every wave walks the 27 offsets in order
    L.check(lib.v3d_sparse_conv_fwd_packed(None, L.ptr(img), L.ptr(nbr_dev), L.ptr(n_dev), n, K, cin, cin, L.ptr(sc), L.ptr(sh), 1, L.ptr(out_a),
                                            -variant, prec, L.ptr(ent_in), L.ptr(ent_next), L.ptr(flag), L.ptr(fsplit), L.ptr(os_a), L.stream_ptr()), "packed2")
    L.check(lib.v3d_sparse_conv_fwd_brick(L.ptr(fsplit), L.ptr(img), L.ptr(nbr_dev), *[L.ptr(t) for t in tabs], L.ptr(n_dev), n, K, cin, cin,
                                          L.ptr(sc), L.ptr(sh), 1, L.ptr(out_b), prec, L.ptr(ent_in), L.ptr(ent_next), L.ptr(flag), L.ptr(os_b),
                                          L.stream_ptr()), "brick")
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b), float((out_a - out_b).abs().max())
    assert torch.equal(os_a, os_b)
    ref = conv64(feats, w, nbr, scale, shift, relu=True)
    got = out_b.cpu().numpy()
    if precision == "fp32":
        assert_features_close(got, ref, f"brick {cin} {order}", floor=FP32_CLASS_FLOOR)
        assert strict_rel_err(got, ref) < STRICT_FP32_CLASS
    else:
        assert_features_close(got, ref, f"brick {cin} {order}")
    # split rows only (what a plan in throughput mode asks for): same split rows
    os_c = torch.zeros_like(os_a)
    L.check(lib.v3d_sparse_conv_fwd_brick(L.ptr(fsplit), L.ptr(img), L.ptr(nbr_dev), *[L.ptr(t) for t in tabs], L.ptr(n_dev), n, K, cin, cin,
                                          L.ptr(sc), L.ptr(sh), 1, None, prec, L.ptr(ent_in), L.ptr(ent_next), L.ptr(flag), L.ptr(os_c),
                                          L.stream_ptr()), "brick")
    torch.cuda.synchronize()
    assert torch.equal(os_c, os_b)


def test_brick_refuses_what_it_does_not_cover():
    from vision3d_amd import _lib as L
    lib = L.lib()
    one = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    n_dev = torch.tensor([16], dtype=torch.int32, device="cuda")
    p = L.ptr(one)
    unsupported = lib.v3d_sparse_brick_plan(p, L.ptr(n_dev), 16, 9, p, p, p, p, L.stream_ptr())
    assert unsupported != 0  # 3x3x3 tables only
    rc = lib.v3d_sparse_conv_fwd_brick(p, p, p, p, p, p, p, L.ptr(n_dev), 16, 27, 16, 16, None, None, 0, p, 0, None, None, None, None, L.stream_ptr())
    assert rc != 0  # 64 -> 64 and 32 -> 32 only
