"""furthest_point_sample / gather_operation / ball_query / grouping_operation / QueryAndGroup
(csrc/pointops.hip, csrc/sa_mlp.hip).  gather_operation and grouping_operation are differentiable in `features`
(upstream pointnet2 implements both backward passes: a scatter-add of the output gradient)."""
import torch
from torch import nn

from .. import _lib as L


def furthest_point_sample(xyz, npoint):
    """xyz (B, N, 3) float32 -> idx (B, npoint) int32; first index 0, ties -> lowest index."""
    L.require_gpu("furthest_point_sample", xyz)
    p = L.as_f32("furthest_point_sample", xyz)
    b, n, _ = p.shape
    idx = torch.empty((b, npoint), dtype=torch.int32, device=p.device)
    with L.device_guard(p.device):
        L.check(L.lib().v3d_furthest_point_sample(L.ptr(p), b, n, int(npoint), L.ptr(idx), 0, 0, L.stream_ptr()),
                "furthest_point_sample")
    return idx


def _scatter_add_backward(grad_out, idx_flat, n):
    """grad_out (B, C, J), idx_flat (B, J) -> (B, C, N): gradient of a gather along the last axis."""
    b, c, _ = grad_out.shape
    grad = grad_out.new_zeros((b, c, n))
    return grad.scatter_add_(2, idx_flat.long().unsqueeze(1).expand(-1, c, -1), grad_out)


class _GatherFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return _gather_forward(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _scatter_add_backward(grad_out.contiguous(), idx, ctx.n), None


class _GroupFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return _group_forward(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        b, c, m, ns = grad_out.shape
        return _scatter_add_backward(grad_out.reshape(b, c, m * ns), idx.reshape(b, m * ns), ctx.n), None


def gather_operation(features, idx):
    """features (B, C, N), idx (B, K) int32 -> (B, C, K); differentiable in `features`."""
    if torch.is_grad_enabled() and features.requires_grad:
        return _GatherFunction.apply(features, idx)
    return _gather_forward(features, idx)


def _gather_forward(features, idx):
    L.require_gpu("gather_operation", features, idx)
    f, i = L.as_f32("gather_operation", features.detach()), L.as_i32("gather_operation", idx)
    b, c, n = f.shape
    k = i.shape[1]
    out = torch.empty((b, c, k), dtype=torch.float32, device=f.device)
    with L.device_guard(f.device):
        L.check(L.lib().v3d_gather_points(L.ptr(f), L.ptr(i), b, c, n, k, L.ptr(out), L.stream_ptr()), "gather_points")
    return out


# "grid" (default): csrc/pointops.hip v3d_ball_query_grid -- the database binned into (x, y) cells, a query tests the 3 x 3 cells
# around its own; "scan": v3d_ball_query -- every database point against every query.  Same results bit for bit (the tests compare
# them); the switch exists for A/B measurements.
BALL_QUERY_ALGO = "grid"


class BallQueryGrid:
    """The cell grid of one database (csrc/pointops.hip: records sorted by (index chunk, (x, y) cell)), built for radii up to
    `radius_max`; holds its workspace.  `matches(xyz)`: was it built from this tensor as it is now."""

    def __init__(self, xyz, radius_max, workspace):
        self.workspace, self.radius_max = workspace, float(radius_max)
        self.shape, self.stamp = tuple(xyz.shape), (xyz.data_ptr(), xyz._version)

    def matches(self, xyz, radius):
        return (tuple(xyz.shape) == self.shape and (xyz.data_ptr(), xyz._version) == self.stamp
                and abs(float(radius)) <= self.radius_max)


def ball_query_grids(databases):
    """[(xyz (B, N, 3) float32 contiguous, radius_max)] (same B) -> [BallQueryGrid], ALL grids in one launch (a workgroup per database
    and frame): the six databases of a PV-RCNN frame cost one build instead of six."""
    import ctypes as C
    out = []
    for i in range(0, len(databases), 8):
        part = databases[i:i + 8]
        for xyz, r in part:
            L.require_gpu("ball_query_grids", xyz)
            if xyz.dtype != torch.float32 or not xyz.is_contiguous() or xyz.dim() != 3 or xyz.shape[2] != 3 or xyz.shape[0] != part[0][0].shape[0]:
                raise RuntimeError("ball_query_grids: databases must be contiguous float32 (B, N, 3) with one B")
            if not abs(float(r)) > 0:
                raise RuntimeError("ball_query_grids: radius_max must be non-zero")
        b, n_db = part[0][0].shape[0], len(part)
        dev = part[0][0].device
        sizes = [int(L.lib().v3d_ball_query_grid_workspace(b, xyz.shape[1])) for xyz, _ in part]
        wss = [torch.empty(max(sz, 16), dtype=torch.uint8, device=dev) for sz in sizes]
        with L.device_guard(dev):
            L.check(L.lib().v3d_ball_query_grid_build(
                n_db, (C.c_void_p * n_db)(*[xyz.data_ptr() for xyz, _ in part]), (C.c_int32 * n_db)(*[xyz.shape[1] for xyz, _ in part]),
                (C.c_float * n_db)(*[abs(float(r)) for _, r in part]), (C.c_void_p * n_db)(*[w.data_ptr() for w in wss]),
                (C.c_size_t * n_db)(*[w.numel() for w in wss]), b, L.stream_ptr()), "ball_query_grid_build")
        out += [BallQueryGrid(xyz, abs(float(r)), w) for (xyz, r), w in zip(part, wss)]
    return out


def _ball_query_call(p, q, radius_a, nsample_a, idx_a, radius_b, nsample_b, idx_b, what, grid=None):
    b, n, _ = p.shape
    m = q.shape[1]
    with L.device_guard(p.device):
        if grid is not None and BALL_QUERY_ALGO == "grid":
            if not grid.matches(p, max(abs(radius_a), abs(radius_b) if idx_b is not None else 0.0)):
                raise RuntimeError(f"{what}: the grid was built from another database (or for a smaller radius)")
            L.check(L.lib().v3d_ball_query_grid_query(L.ptr(q), b, n, m, float(radius_a), int(nsample_a), L.ptr(idx_a), float(radius_b),
                                                      int(nsample_b), L.ptr(idx_b), L.ptr(grid.workspace), grid.workspace.numel(),
                                                      L.stream_ptr()), what)
        elif BALL_QUERY_ALGO == "grid":
            nbytes = L.lib().v3d_ball_query_grid_workspace(b, n)
            ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=p.device)
            L.check(L.lib().v3d_ball_query_grid(L.ptr(p), L.ptr(q), b, n, m, float(radius_a), int(nsample_a), L.ptr(idx_a),
                                                float(radius_b), int(nsample_b), L.ptr(idx_b), L.ptr(ws), ws.numel(), L.stream_ptr()),
                    what)
        elif BALL_QUERY_ALGO == "scan":
            L.check(L.lib().v3d_ball_query(L.ptr(p), L.ptr(q), b, n, m, float(radius_a), int(nsample_a), L.ptr(idx_a), float(radius_b),
                                           int(nsample_b), L.ptr(idx_b), L.stream_ptr()), what)
        else:
            raise ValueError(f"BALL_QUERY_ALGO must be 'grid' or 'scan', not {BALL_QUERY_ALGO!r}")


def ball_query(radius, nsample, xyz, new_xyz, grid=None):
    """xyz (B, N, 3), new_xyz (B, M, 3) -> idx (B, M, nsample) int32: the first `nsample` points (index
    order) with d^2 < r^2, empty slots filled with the first hit, no hit -> 0.  `grid`: a BallQueryGrid of `xyz` built beforehand
    (`ball_query_grids`); default: built here."""
    L.require_gpu("ball_query", xyz, new_xyz)
    p, q = L.as_f32("ball_query", xyz), L.as_f32("ball_query", new_xyz)
    idx = torch.empty((p.shape[0], q.shape[1], nsample), dtype=torch.int32, device=p.device)
    _ball_query_call(p, q, radius, nsample, idx, 0.0, 0, None, "ball_query", grid)
    return idx


def ball_query_pair(radius_a, nsample_a, radius_b, nsample_b, xyz, new_xyz, grid=None):
    """Two ball queries around the same `new_xyz` in one pass over `xyz` (the two scales of a multi-scale set-abstraction module):
    -> (idx_a (B, M, nsample_a), idx_b (B, M, nsample_b)), each exactly what `ball_query` returns for its radius."""
    L.require_gpu("ball_query", xyz, new_xyz)
    p, q = L.as_f32("ball_query", xyz), L.as_f32("ball_query", new_xyz)
    idx_a = torch.empty((p.shape[0], q.shape[1], nsample_a), dtype=torch.int32, device=p.device)
    idx_b = torch.empty((p.shape[0], q.shape[1], nsample_b), dtype=torch.int32, device=p.device)
    _ball_query_call(p, q, radius_a, nsample_a, idx_a, radius_b, nsample_b, idx_b, "ball_query2", grid)
    return idx_a, idx_b


def ball_query_pairs_many(jobs, new_xyz):
    """Several two-radius ball queries around the SAME new_xyz (B, M, 3), each in its own database, in ONE launch
    (v3d_ball_query_grid_query_many): jobs = [(grid: BallQueryGrid, xyz, radius_a, nsample_a, radius_b, nsample_b)] (at most 8)
    -> [(idx_a, idx_b)], each pair exactly what `ball_query_pair` returns."""
    import ctypes as C
    L.require_gpu("ball_query_many", new_xyz)
    q = L.as_f32("ball_query_many", new_xyz)
    b, m, _ = q.shape
    n_jobs = len(jobs)
    outs = []
    for grid, xyz, ra, nsa, rb, nsb in jobs:
        if not grid.matches(xyz, max(abs(ra), abs(rb))) or xyz.shape[0] != b:
            raise RuntimeError("ball_query_many: a grid was built from another database (or for a smaller radius)")
        outs.append((torch.empty((b, m, nsa), dtype=torch.int32, device=q.device), torch.empty((b, m, nsb), dtype=torch.int32, device=q.device)))
    vp, i32, f32 = C.c_void_p * n_jobs, C.c_int32 * n_jobs, C.c_float * n_jobs
    with L.device_guard(q.device):
        L.check(L.lib().v3d_ball_query_grid_query_many(
            n_jobs, L.ptr(q), b, m, i32(*[j[1].shape[1] for j in jobs]), f32(*[float(j[2]) for j in jobs]), i32(*[int(j[3]) for j in jobs]),
            vp(*[o[0].data_ptr() for o in outs]), f32(*[float(j[4]) for j in jobs]), i32(*[int(j[5]) for j in jobs]),
            vp(*[o[1].data_ptr() for o in outs]), vp(*[j[0].workspace.data_ptr() for j in jobs]),
            (C.c_size_t * n_jobs)(*[j[0].workspace.numel() for j in jobs]), L.stream_ptr()), "ball_query_grid_query_many")
    return outs


def linear_rows_many(jobs):
    """[(a (R, K), w (K, Nout))] (at most 8) -> [a @ w], all products in ONE launch (v3d_linear_rows_many; no bias, no ReLU: the
    first-layer feature products of the set-abstraction modules of a frame)."""
    import ctypes as C
    n_jobs = len(jobs)
    outs = []
    for a, w in jobs:
        L.require_gpu("linear_rows_many", a, w)
        if a.dtype != torch.float32 or a.dim() != 2 or a.stride(1) != 1 or w.dtype != torch.float32 or not w.is_contiguous() or w.shape[0] != a.shape[1]:
            raise RuntimeError("linear_rows_many: (R, K) float32 rows with contiguous columns and contiguous (K, Nout) weights")
        outs.append(torch.empty((a.shape[0], w.shape[1]), dtype=torch.float32, device=a.device))
    vp, i32 = C.c_void_p * n_jobs, C.c_int32 * n_jobs
    with L.device_guard(jobs[0][0].device):
        L.check(L.lib().v3d_linear_rows_many(
            n_jobs, vp(*[a.data_ptr() for a, _ in jobs]), i32(*[a.stride(0) if a.shape[0] > 1 else max(a.stride(0), a.shape[1]) for a, _ in jobs]),
            i32(*[a.shape[0] for a, _ in jobs]), i32(*[a.shape[1] for a, _ in jobs]), vp(*[w.data_ptr() for _, w in jobs]), None,
            i32(*[w.shape[1] for _, w in jobs]), None, vp(*[o.data_ptr() for o in outs]), None, None, L.stream_ptr()), "linear_rows_many")
    return outs


def grouping_operation(features, idx):
    """features (B, C, N), idx (B, M, ns) int32 -> (B, C, M, ns); differentiable in `features`."""
    if torch.is_grad_enabled() and features.requires_grad:
        return _GroupFunction.apply(features, idx)
    return _group_forward(features, idx)


def _group_forward(features, idx):
    L.require_gpu("grouping_operation", features, idx)
    f, i = L.as_f32("grouping_operation", features.detach()), L.as_i32("grouping_operation", idx)
    b, c, n = f.shape
    _, m, ns = i.shape
    out = torch.empty((b, c, m, ns), dtype=torch.float32, device=f.device)
    with L.device_guard(f.device):
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


def _strided_rows(name, out, rows, cols):
    """(rows, >= cols) float32 view with unit column stride -> (tensor, row stride): a column block of a wider matrix."""
    if (out.dtype != torch.float32 or out.dim() != 2 or out.shape[0] != rows or out.shape[1] < cols or out.stride(1) != 1
            or out.stride(0) < cols):
        raise RuntimeError(f"{name}: `out` must be a ({rows}, >= {cols}) float32 view with contiguous columns")
    return out, out.stride(0)


def sa_mlp_layer(feat, w, bias, relu=True, pool=False, xyz=None, new_xyz=None, idx=None, groups=None, out=None, n_store=None):
    """One layer of a set-abstraction shared MLP on the matrix cores (csrc/sa_mlp.hip, exact fp32 MFMA).

    first layer:  feat (B, N, Kf) point-major, xyz (B, N, 3), new_xyz (B, M, 3), idx (B, M, ns) int32;
                  row (b, m, s) = [xyz[i] - new_xyz[m], 0 | feat[i]] with i = idx[b, m, s], w is (4 + Kf, Nout)
    later layers: feat (B*M*ns, Kf) (the previous layer's output), groups = (B, M, ns), w is (Kf, Nout)
    Kf % 4 == 0, Nout % 16 == 0.  Returns (B*M*ns, Nout), or (B*M, Nout) = max over the ns rows of a group with pool=True.
    `out`: a (rows, >= n_store) view with unit column stride to write into (e.g. a column block of the keypoint feature matrix);
    `n_store`: only the first n_store output columns are stored (default Nout)."""
    L.require_gpu("sa_mlp_layer", feat, w)
    f, wf = L.as_f32("sa_mlp_layer", feat), L.as_f32("sa_mlp_layer", w)
    bf = None if bias is None else L.as_f32("sa_mlp_layer", bias)
    if idx is not None:
        b, n, kf = f.shape
        _, m, ns = idx.shape
        x, q, ii = L.as_f32("sa_mlp_layer", xyz), L.as_f32("sa_mlp_layer", new_xyz), L.as_i32("sa_mlp_layer", idx)
    else:
        b, m, ns = groups
        n, kf = m * ns, f.shape[1]
        x = q = ii = None
        if f.shape[0] != b * m * ns:
            raise RuntimeError("sa_mlp_layer: rows do not match (B, M, ns)")
    nout = wf.shape[1]
    if wf.shape[0] != kf + (4 if idx is not None else 0):
        raise RuntimeError("sa_mlp_layer: weight rows do not match the input width")
    rows = b * m if pool else b * m * ns
    cols = nout if n_store is None else int(n_store)
    if out is None:
        out, ldo = torch.empty((rows, cols), dtype=torch.float32, device=f.device), cols
    else:
        out, ldo = _strided_rows("sa_mlp_layer", out, rows, cols)
    with L.device_guard(f.device):
        L.check(L.lib().v3d_sa_mlp_layer(L.ptr(f), L.ptr(x), L.ptr(q), L.ptr(ii), b, n, m, ns, kf, L.ptr(wf), L.ptr(bf), nout,
                                         int(bool(relu)), int(bool(pool)), L.ptr(out), ldo, cols, L.stream_ptr()), "sa_mlp_layer")
    return out


def sa_mlp_pair(p, xyz, new_xyz, idx, wx, b1, w, bias, relu=True, pool=True, out=None, n_store=None):
    """The first two layers of a set-abstraction scale in one launch (csrc/sa_mlp.hip PAIR): p (B, N, K1) = feat @ W1[4:] (the first
    layer's feature part, once per database point: `linear_rows`; may be a column block of a wider (B, N, >= K1) matrix), wx (3, K1)
    = W1[0:3], b1 (K1); the kernel rebuilds relu(p[i] + rel_xyz . wx + b1) for every grouped row and multiplies it by w (K1, Nout)
    (+ bias, ReLU, max over the samples)."""
    L.require_gpu("sa_mlp_pair", p, w)
    wf = L.as_f32("sa_mlp_pair", w)
    x, q, ii = L.as_f32("sa_mlp_pair", xyz), L.as_f32("sa_mlp_pair", new_xyz), L.as_i32("sa_mlp_pair", idx)
    wxf, b1f = L.as_f32("sa_mlp_pair", wx), L.as_f32("sa_mlp_pair", b1)
    bf = None if bias is None else L.as_f32("sa_mlp_pair", bias)
    b, n, k1 = p.shape
    if p.dtype != torch.float32 or p.stride(2) != 1 or p.stride(0) != n * p.stride(1):
        raise RuntimeError("sa_mlp_pair: p must be a float32 (B, N, K1) view with unit channel stride and frames back to back")
    _, m, ns = ii.shape
    nout = wf.shape[1]
    if wf.shape[0] != k1 or tuple(wxf.shape) != (3, k1) or b1f.numel() != k1:
        raise RuntimeError("sa_mlp_pair: weight shapes do not match the first layer's width")
    rows = b * m if pool else b * m * ns
    cols = nout if n_store is None else int(n_store)
    if out is None:
        out, ldo = torch.empty((rows, cols), dtype=torch.float32, device=p.device), cols
    else:
        out, ldo = _strided_rows("sa_mlp_pair", out, rows, cols)
    with L.device_guard(p.device):
        L.check(L.lib().v3d_sa_mlp_pair(L.ptr(p), L.ptr(x), L.ptr(q), L.ptr(ii), b, n, m, ns, k1, p.stride(1), L.ptr(wxf), L.ptr(b1f),
                                        L.ptr(wf), L.ptr(bf), nout, int(bool(relu)), int(bool(pool)), L.ptr(out), ldo, cols,
                                        L.stream_ptr()), "sa_mlp_pair")
    return out


def sa_mlp_pair2(p_a, p_b, xyz, new_xyz, idx_a, idx_b, wx_a, b1_a, wx_b, b1_b, w_a, bias_a, w_b, bias_b, out_a, out_b, n_store):
    """`sa_mlp_pair` (ReLU, max over the samples) for the TWO scales of a module in one launch: p_a / p_b (B, N, K1) column blocks of one
    product, the same K1 and Nout, out_a / out_b (B*M, >= n_store) column blocks of one matrix (the same row stride)."""
    L.require_gpu("sa_mlp_pair2", p_a, w_a)
    x, q = L.as_f32("sa_mlp_pair2", xyz), L.as_f32("sa_mlp_pair2", new_xyz)
    ia, ib = L.as_i32("sa_mlp_pair2", idx_a), L.as_i32("sa_mlp_pair2", idx_b)
    b, n, k1 = p_a.shape
    _, m, ns_a = ia.shape
    ns_b = ib.shape[2]
    nout = w_a.shape[1]
    for p in (p_a, p_b):
        if p.dtype != torch.float32 or tuple(p.shape) != (b, n, k1) or p.stride(2) != 1 or p.stride(0) != n * p.stride(1) or p.stride(1) != p_a.stride(1):
            raise RuntimeError("sa_mlp_pair2: p_a / p_b must be (B, N, K1) column blocks of one float32 matrix")
    if tuple(w_a.shape) != (k1, nout) or tuple(w_b.shape) != (k1, nout) or tuple(wx_a.shape) != (3, k1) or tuple(wx_b.shape) != (3, k1):
        raise RuntimeError("sa_mlp_pair2: the two scales must have the same widths")
    oa, ldo = _strided_rows("sa_mlp_pair2", out_a, b * m, n_store)
    ob, ldo_b = _strided_rows("sa_mlp_pair2", out_b, b * m, n_store)
    if ldo != ldo_b:
        raise RuntimeError("sa_mlp_pair2: out_a / out_b must be column blocks of one matrix")
    with L.device_guard(p_a.device):
        L.check(L.lib().v3d_sa_mlp_pair2(L.ptr(p_a), L.ptr(p_b), L.ptr(x), L.ptr(q), L.ptr(ia), L.ptr(ib), b, n, m, ns_a, ns_b, k1, p_a.stride(1),
                                         L.ptr(wx_a), L.ptr(b1_a), L.ptr(wx_b), L.ptr(b1_b), L.ptr(w_a), L.ptr(bias_a), L.ptr(w_b), L.ptr(bias_b),
                                         nout, 1, 1, L.ptr(oa), L.ptr(ob), ldo, int(n_store), L.stream_ptr()), "sa_mlp_pair2")


def linear_rows(a, w, bias=None, relu=False, out=None, n_store=None):
    """act(a @ w + bias) for a matrix of FEW rows (csrc/sa_mlp.hip linear_rows_kernel: columns over workgroups, K over the waves):
    a (R, K) float32 with unit column stride (rows may be strided), w (K, Nout) = the nn.Linear weight transposed, Nout % 16 == 0,
    K % 4 == 0.  -> (R, n_store) (default Nout), or written into `out` (a (R, >= n_store) view with unit column stride)."""
    L.require_gpu("linear_rows", a, w)
    wf = L.as_f32("linear_rows", w)
    bf = None if bias is None else L.as_f32("linear_rows", bias)
    if a.dtype != torch.float32 or a.dim() != 2 or a.stride(1) != 1:
        raise RuntimeError("linear_rows: `a` must be a 2-D float32 matrix with contiguous columns")
    r, k = a.shape
    nout = wf.shape[1]
    if wf.shape[0] != k:
        raise RuntimeError("linear_rows: weight rows do not match the input width")
    cols = nout if n_store is None else int(n_store)
    if out is None:
        out, ldo = torch.empty((r, cols), dtype=torch.float32, device=a.device), cols
    else:
        out, ldo = _strided_rows("linear_rows", out, r, cols)
    with L.device_guard(a.device):
        L.check(L.lib().v3d_linear_rows(L.ptr(a), a.stride(0) if r > 1 else max(a.stride(0), k), r, k, L.ptr(wf), L.ptr(bf), nout,
                                        int(bool(relu)), L.ptr(out), ldo, cols, L.stream_ptr()), "linear_rows")
    return out
