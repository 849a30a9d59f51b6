"""GPU parity: device voxelizer + fused VFE mean vs oracle/ (bit-exact: indices, occupancy, point slots;
the mean is the same fp32 sum in the same order -> also compared exactly)."""
import numpy as np
import pytest
import torch

from gpu_util import dev
from vision3d_amd import synth

pytestmark = pytest.mark.gpu
VS = [0.05, 0.05, 0.1]


def gpu_voxelize(clouds, bounds, max_pts=5, max_voxels=20000):
    from vision3d_amd.spconv.utils import voxelize_batch
    offs = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
    flat = dev(np.concatenate(clouds) if len(clouds) else np.zeros((0, 4), np.float32), torch.float32)
    vox, coords, occ, mean, n = voxelize_batch(flat, offs, VS, bounds, max_pts, max_voxels)
    m = int(n.item())
    return vox[:m].cpu().numpy(), coords[:m].cpu().numpy(), occ[:m].cpu().numpy(), mean[:m].cpu().numpy()


def oracle_voxelize(oracle, clouds, bounds, max_pts=5, max_voxels=20000):
    from oracle import second_cpu
    vox, coords, occ = second_cpu.voxelize_batch(clouds, VS, bounds, max_pts, max_voxels)
    return vox, coords, occ, oracle.vfe_mean(vox, occ)


def compare(got, ref):
    for g, r, name in zip(got, ref, ("voxels", "coords", "occupancy", "mean")):
        assert g.shape == r.shape, (name, g.shape, r.shape)
        np.testing.assert_array_equal(g, r, err_msg=name)


@pytest.mark.parametrize("seed", [0, 1])
def test_single_kitti_frame(oracle, seed):
    cloud = synth.make_cloud(seed)
    compare(gpu_voxelize([cloud], synth.KITTI_BOUNDS), oracle_voxelize(oracle, [cloud], synth.KITTI_BOUNDS))


def test_ragged_batch_edges(oracle):
    clouds = [synth.make_cloud(2)[:9000], synth.make_cloud(3)[:1], synth.make_cloud(4), synth.make_cloud(5)[:12345]]
    clouds[0][::5, 0] = 70.4          # on the upper bound: dropped
    clouds[0][::9, 1] = -40.0         # on the lower bound: kept
    clouds[2][::3] = clouds[2][0]     # heavy duplicates -> occupancy clipping at 5 with first-come slots
    clouds[3][7, 2] = np.nan
    compare(gpu_voxelize(clouds, synth.KITTI_BOUNDS), oracle_voxelize(oracle, clouds, synth.KITTI_BOUNDS))


def test_max_voxels_clip_and_small_max_pts(oracle):
    clouds = [synth.make_cloud(6), synth.make_cloud(7)[:3000], synth.make_cloud(8)]
    compare(gpu_voxelize(clouds, synth.KITTI_BOUNDS, 3, 2500), oracle_voxelize(oracle, clouds, synth.KITTI_BOUNDS, 3, 2500))


def test_empty_and_all_outside():
    from vision3d_amd.spconv.utils import voxelize_batch
    _, coords, _, _, n = voxelize_batch(torch.zeros(0, 4).cuda(), [0, 0], VS, synth.KITTI_BOUNDS, 5, 20000)
    assert int(n.item()) == 0
    far = torch.full((100, 4), 1e6).cuda()
    _, _, _, _, n = voxelize_batch(far, [0, 100], VS, synth.KITTI_BOUNDS, 5, 20000)
    assert int(n.item()) == 0


def test_voxel_generator_drop_in(oracle):
    """spconv.utils.VoxelGenerator.generate contract: numpy in -> numpy (voxels, zyx coords, counts)."""
    from vision3d_amd.spconv.utils import VoxelGenerator
    cloud = synth.make_cloud(9)
    gen = VoxelGenerator(voxel_size=VS, point_cloud_range=list(synth.KITTI_BOUNDS), max_voxels=20000, max_num_points=5)
    v, c, n = gen.generate(cloud)
    rv, rc, rn = oracle.voxelize(cloud, VS, synth.KITTI_BOUNDS, 5, 20000)
    assert isinstance(v, np.ndarray) and c.dtype == np.int32 and list(gen.grid_size) == [1408, 1600, 40]
    np.testing.assert_array_equal(v, rv); np.testing.assert_array_equal(c, rc); np.testing.assert_array_equal(n, rn)


def test_waymo_range_full_size_properties():
    """BASELINE configs[4] size (180k points, 0.55 G-cell grid): size-independent properties instead of
    the slow oracle -- voxel set == np.unique of fp32 floor coordinates, first-touch order, occupancy."""
    cloud = synth.make_waymo_cloud(0)
    vox, coords, occ, mean = gpu_voxelize([cloud], synth.WAYMO_BOUNDS, 5, 400000)
    lo = np.array(synth.WAYMO_BOUNDS[:3], np.float32)
    c = np.floor((cloud[:, :3] - lo) / np.array(VS, np.float32)).astype(np.int64)
    grid = np.array([3008, 3008, 60])
    ok = np.all((c >= 0) & (c < grid), 1)
    uniq, first, counts = np.unique(c[ok][:, ::-1], axis=0, return_index=True, return_counts=True)
    order = np.argsort(first)
    np.testing.assert_array_equal(coords[:, 1:], uniq[order])
    np.testing.assert_array_equal(occ, np.minimum(counts[order], 5))
    np.testing.assert_array_equal(vox[:, 0], cloud[ok][first[order]])
    assert (coords[:, 0] == 0).all() and len(coords) > 90000
