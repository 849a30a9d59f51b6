"""ctypes/numpy front-end of the CPU checker (oracle/v3d_oracle.c and oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under vision3d_amd/ imports this module.

Pinning: IoU/NMS pinned to the reference (oracle/_ref + tests/golden); voxelizer / sparse conv /
pointnet2 ops are "parity unpinned" (third-party sources absent from /root/reference) -- see
v3d_oracle.c header and DESIGN.md.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "_build", "libv3d_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libv3d_ref.so")
REF_ROOT = "/root/reference"


def build(force=False):
    """Compile the plain-C restatement (always) and oracle/_ref (only when /root/reference exists)."""
    targets = ["all"]
    if os.path.isdir(REF_ROOT):
        targets.append("ref")
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE] + targets, stdout=subprocess.DEVNULL)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_ORACLE_SO):
            build()
        _lib = C.CDLL(_ORACLE_SO)
        _lib.orc_single_box_iou_rotated.restype = C.c_float
        _lib.orc_nms_margin.restype = C.c_float
    return _lib


def have_ref():
    return os.path.exists(_REF_SO)


def ref():
    """The reference's own IoU core (oracle/_ref/libv3d_ref.so), if it has been built."""
    global _ref
    if _ref is None:
        if not have_ref():
            raise FileNotFoundError(_REF_SO + " (run `make -C oracle ref` where /root/reference exists)")
        _ref = C.CDLL(_REF_SO)
        _ref.ref_single_box_iou_rotated.restype = C.c_float
    return _ref


# ---------------------------------------------------------------- rotated IoU / NMS
def box_iou_rotated(b1, b2, use_ref=False):
    b1, b2 = _f32(b1).reshape(-1, 5), _f32(b2).reshape(-1, 5)
    out = np.empty((b1.shape[0], b2.shape[0]), np.float32)
    fn = ref().ref_box_iou_rotated if use_ref else lib().orc_box_iou_rotated
    fn(_fp(b1), b1.shape[0], _fp(b2), b2.shape[0], _fp(out))
    return out


def box_iou_rotated_3d(b1, b2):
    """(M, 7) x (N, 7) -> (M, N): BEV intersection (box_iou_rotated's operator on columns 0,1,3,4,6) x z overlap / union volume."""
    b1, b2 = _f32(b1).reshape(-1, 7), _f32(b2).reshape(-1, 7)
    out = np.empty((b1.shape[0], b2.shape[0]), np.float32)
    lib().orc_box_iou_rotated_3d(_fp(b1), b1.shape[0], _fp(b2), b2.shape[0], _fp(out))
    return out


def score_order(scores):
    """Descending-score permutation; ties -> lower index first (stable)."""
    return np.argsort(-_f32(scores), kind="stable").astype(np.int64)


def nms_rotated(boxes, scores, thr, use_ref=False):
    boxes = _f32(boxes).reshape(-1, 5)
    n = boxes.shape[0]
    order = score_order(scores)
    keep = np.empty(max(n, 1), np.int64)
    fn = ref().ref_nms_rotated if use_ref else lib().orc_nms_rotated
    nk = fn(_fp(boxes), _lp(order), n, C.c_float(thr), _lp(keep))
    return keep[:nk].copy()


def nms_margin(boxes, scores, thr):
    boxes = _f32(boxes).reshape(-1, 5)
    order = score_order(scores)
    return float(lib().orc_nms_margin(_fp(boxes), _lp(order), boxes.shape[0], C.c_float(thr)))


def batched_nms_rotated(boxes, scores, idxs, thr, use_ref=False):
    """vision3d/ops/iou_nms.py:90-134 (coordinate-offset trick), numpy restatement."""
    boxes = _f32(boxes).reshape(-1, 5)
    if boxes.size == 0:
        return np.empty((0,), np.int64)
    mx = (np.maximum(boxes[:, 0], boxes[:, 1]) + np.maximum(boxes[:, 2], boxes[:, 3]) / np.float32(2)).max()
    mn = (np.minimum(boxes[:, 0], boxes[:, 1]) - np.minimum(boxes[:, 2], boxes[:, 3]) / np.float32(2)).min()
    offsets = np.asarray(idxs).astype(np.float32) * (mx - mn + np.float32(1))
    b = boxes.copy()
    b[:, :2] += offsets[:, None]
    return nms_rotated(b, scores, thr, use_ref)


# ---------------------------------------------------------------- voxelizer / VFE
def voxelize(points, voxel_size, bounds, max_pts, max_voxels):
    """spconv VoxelGenerator.generate restatement -> (voxels (M,max_pts,C), coors (M,3) zyx, num (M,))."""
    points = _f32(points)
    n, c = points.shape
    vs, rg = _f32(voxel_size), _f32(bounds)
    voxels = np.empty((max_voxels, max_pts, c), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.empty((max_voxels,), np.int32)
    m = lib().orc_voxelize(_fp(points), n, c, _fp(vs), _fp(rg), max_pts, max_voxels, _fp(voxels), _ip(coors), _ip(num))
    return voxels[:m].copy(), coors[:m].copy(), num[:m].copy()


def vfe_mean(voxels, num):
    voxels, num = _f32(voxels), _i32(num)
    m, k, c = voxels.shape
    out = np.empty((m, c), np.float32)
    lib().orc_vfe_mean(_fp(voxels), _ip(num), m, k, c, _fp(out))
    return out


# ---------------------------------------------------------------- sparse conv
def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def _triple(v):
    return [int(v)] * 3 if np.isscalar(v) else [int(x) for x in v]


def subm_rulebook(coords, shape, ksize=3):
    coords = _i32(coords).reshape(-1, 4)
    ks = _triple(ksize)
    k = ks[0] * ks[1] * ks[2]
    nbr = np.empty((coords.shape[0], k), np.int32)
    lib().orc_subm_rulebook(_ip(coords), coords.shape[0], _i3(shape), _i3(ks), _ip(nbr))
    return nbr


def sparse_rulebook(coords, shape, ksize, stride, padding):
    coords = _i32(coords).reshape(-1, 4)
    ks, st, pd = _triple(ksize), _triple(stride), _triple(padding)
    k = ks[0] * ks[1] * ks[2]
    n = coords.shape[0]
    out_shape = _i3([0, 0, 0])
    max_out = max(8 * n, 1)
    out_coords = np.empty((max_out, 4), np.int32)
    nbr = np.empty((max_out, k), np.int32)
    n_out = lib().orc_sparse_rulebook(_ip(coords), n, _i3(shape), _i3(ks), _i3(st), _i3(pd), _ip(out_coords),
                                      _ip(nbr), max_out, out_shape)
    assert n_out >= 0
    return out_coords[:n_out].copy(), nbr[:n_out].copy(), list(out_shape)


def sparse_conv_fwd(feat, weight, nbr, scale=None, shift=None, relu=False):
    """feat (N_in,Cin), weight (K,Cin,Cout) or (k0,k1,k2,Cin,Cout), nbr (N_out,K) -> (N_out,Cout)."""
    feat, nbr = _f32(feat), _i32(nbr)
    w = _f32(weight)
    cin, cout = w.shape[-2], w.shape[-1]
    w = w.reshape(-1, cin, cout)
    n_out, k = nbr.shape
    assert w.shape[0] == k and feat.shape[1] == cin
    out = np.empty((n_out, cout), np.float32)
    sc = sh = None
    if scale is not None:
        scale_a, shift_a = _f32(scale), _f32(shift)  # keep the arrays alive across the call
        sc, sh = _fp(scale_a), _fp(shift_a)
    lib().orc_sparse_conv_fwd(_fp(feat), _fp(w), _ip(nbr), n_out, k, cin, cout, sc, sh, int(bool(relu)), _fp(out))
    return out


def densify(feat, coords, batch_size, shape):
    feat, coords = _f32(feat), _i32(coords).reshape(-1, 4)
    c = feat.shape[1]
    dense = np.empty((batch_size, c, shape[0], shape[1], shape[2]), np.float32)
    lib().orc_densify(_fp(feat), _ip(coords), feat.shape[0], batch_size, c, _i3(shape), _fp(dense))
    return dense


# ---------------------------------------------------------------- pointnet2 ops
def fps(xyz, k):
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    idx = np.empty((b, k), np.int32)
    lib().orc_fps(_fp(xyz), b, n, k, _ip(idx))
    return idx


def gather(feat, idx):
    feat, idx = _f32(feat), _i32(idx)
    b, c, n = feat.shape
    k = idx.shape[1]
    out = np.empty((b, c, k), np.float32)
    lib().orc_gather(_fp(feat), _ip(idx), b, c, n, k, _fp(out))
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    lib().orc_ball_query(_fp(xyz), _fp(new_xyz), b, n, m, C.c_float(radius), nsample, _ip(idx))
    return idx


def group(feat, idx):
    feat, idx = _f32(feat), _i32(idx)
    b, c, n = feat.shape
    _, m, ns = idx.shape
    out = np.empty((b, c, m, ns), np.float32)
    lib().orc_group(_fp(feat), _ip(idx), b, c, n, m, ns, _fp(out))
    return out


# ---------------------------------------------------------------- points in boxes
def points_in_boxes(points, boxes, use_z=True):
    """core/geometry.py PointsInCuboids._get_mask (use_z) / PointsNotInRectangles._get_mask -> (N,n) bool."""
    points, boxes = _f32(points), _f32(boxes).reshape(-1, 7)
    n, c = points.shape
    nb = boxes.shape[0]
    mask = np.empty((n, nb), np.uint8)
    lib().orc_points_in_boxes(_fp(points), n, c, _fp(boxes), nb, int(bool(use_z)),
                              mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    return mask.astype(bool)
