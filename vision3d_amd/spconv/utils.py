"""spconv.utils.VoxelGenerator on the device (call site vision3d/core/preprocess.py:17-33)."""
import numpy as np
import torch

from .. import _lib as L


def voxelize_batch(points, frame_offsets, voxel_size, bounds, max_pts, max_voxels, want_voxels=True):
    """Voxelise B concatenated frames in one pass (csrc/voxelize.hip).

    points (sum N_b, C) float32 cuda; frame_offsets: host list of B+1 ints.  Returns device tensors
    (voxels (M,max_pts,C) | None, coords (M,4) int32 (b,z,y,x), occupancy (M,) int32, mean (M,C)) sized
    by the capacity B*max_voxels, plus n_voxels (1,) int32 on the device -- no host sync here."""
    L.require_gpu("voxelize", points)
    pts = L.as_f32("voxelize", points)
    n, c = pts.shape
    b = len(frame_offsets) - 1
    cap = max(1, min(n, b * int(max_voxels)))
    dev = pts.device
    voxels = torch.empty((cap, max_pts, c), dtype=torch.float32, device=dev) if want_voxels else None
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    occupancy = torch.empty((cap,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap, c), dtype=torch.float32, device=dev)
    n_vox = torch.zeros((1,), dtype=torch.int32, device=dev)
    lib = L.lib()
    ws = L.workspace(lib.v3d_voxelize_workspace(n), dev)
    with torch.cuda.device(dev):
        L.check(lib.v3d_voxelize(L.ptr(pts), n, c, L.host_i32(frame_offsets), b, L.host_f32(voxel_size),
                                 L.host_f32(bounds), int(max_pts), int(max_voxels), L.ptr(voxels), L.ptr(coords),
                                 L.ptr(occupancy), L.ptr(mean), L.ptr(n_vox), L.ptr(ws), ws.numel(), L.stream_ptr()),
                "voxelize")
    return voxels, coords, occupancy, mean, n_vox


class VoxelGenerator(object):
    """Same constructor/`generate` contract as spconv.utils.VoxelGenerator: returns
    (voxels (M,max_num_points,C), coordinates (M,3) int32 zyx, num_points_per_voxel (M,) int32).
    numpy in -> numpy out (drop-in for preprocess.py:30); cuda tensor in -> cuda tensors out."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        self._voxel_size = np.asarray(voxel_size, dtype=np.float32)
        self._point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        grid = (self._point_cloud_range[3:] - self._point_cloud_range[:3]) / self._voxel_size
        self._grid_size = np.round(grid).astype(np.int64)
        self._max_num_points = int(max_num_points)
        self._max_voxels = int(max_voxels)

    voxel_size = property(lambda self: self._voxel_size)
    point_cloud_range = property(lambda self: self._point_cloud_range)
    grid_size = property(lambda self: self._grid_size)
    max_num_points_per_voxel = property(lambda self: self._max_num_points)

    def generate(self, points, max_voxels=None):
        as_numpy = isinstance(points, np.ndarray)
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).cuda() if as_numpy else points
        mv = self._max_voxels if max_voxels is None else int(max_voxels)
        voxels, coords, occ, _, n_vox = voxelize_batch(pts, [0, pts.shape[0]], self._voxel_size,
                                                       self._point_cloud_range, self._max_num_points, mv)
        m = int(n_vox.item())
        out = (voxels[:m], coords[:m, 1:].contiguous(), occ[:m])
        if as_numpy:
            return tuple(t.cpu().numpy() for t in out)
        return out
