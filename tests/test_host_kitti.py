"""CPU: the KITTI file readers (vision3d_amd/dataset/kitti.py) against vectors captured from the
reference's own kitti_utils.py / kitti_dataset.py on synthetic label, calibration and velodyne files
(tests/golden/make_golden_kitti.py; the npz holds the file CONTENTS as data).  Exact comparison: same numpy expressions
in the same dtypes."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kitti():
    from vision3d_amd.dataset import kitti as mod  # host numpy only: importable without a GPU
    return mod


@pytest.fixture(scope="module")
def golden_kitti():
    return np.load(os.path.join(HERE, "golden", "kitti.npz"))


@pytest.mark.parametrize("case", [0, 1, 2])
def test_readers_match_reference(kitti, golden_kitti, tmp_path, case):
    g, k = golden_kitti, f"c{case}_"
    root = tmp_path
    for sub in ("label_2", "calib", "velodyne"):
        os.makedirs(root / sub)
    (root / "label_2" / "000007.txt").write_text(str(g[k + "label_txt"]))
    (root / "calib" / "000007.txt").write_text(str(g[k + "calib_txt"]))
    g[k + "points"].tofile(root / "velodyne" / "000007.bin")

    calib = kitti.read_calib(root / "calib" / "000007.txt")
    for f in ("V2C", "C2V", "R0", "P2", "WH"):
        got = np.asarray(getattr(calib, f))
        assert got.dtype == g[k + "calib_" + f].dtype and np.array_equal(got, g[k + "calib_" + f]), f
    labels = kitti.read_labels(root / "label_2" / "000007.txt")
    np.testing.assert_array_equal(labels.class_idx, g[k + "class_idx"])
    np.testing.assert_array_equal(labels.level, g[k + "level"])
    np.testing.assert_array_equal(labels.location, g[k + "t"])
    np.testing.assert_array_equal(labels.hwl, g[k + "hwl"])
    np.testing.assert_array_equal(labels.box2d, g[k + "box2d"])
    misc = np.stack((labels.truncation, labels.occlusion.astype(np.float64), labels.alpha, labels.ry, labels.score), 1)
    np.testing.assert_array_equal(misc, g[k + "misc"])
    np.testing.assert_array_equal(kitti.boxes_in_lidar_frame(labels, calib), g[k + "boxes"])
    pts = kitti.read_points(root / "velodyne" / "000007.bin")
    np.testing.assert_array_equal(pts, g[k + "points"])
    fov = kitti.crop_to_camera_view(calib, pts)
    assert fov.dtype == np.float32
    np.testing.assert_array_equal(fov, g[k + "fov_points"])
    frame = kitti.load_frame(root, 7, reduced=False)
    np.testing.assert_array_equal(frame["boxes"], g[k + "boxes"])
    assert frame["points"].shape == pts.shape and frame["class_idx"].shape == (len(labels.names),)
