"""KITTI object-detection files either side of the hot path (reference:
vision3d/dataset/kitti_utils.py and the conversions of kitti_dataset.py:74-86).  Host code, numpy only -- this is
on-disk format handling, done once per frame or once per dataset, not device work.

    velodyne/<idx>.bin    float32 (N, 4)  x, y, z, reflectance                      -> read_points
    label_2/<idx>.txt     one object per line, 15 (or 16 with a score) fields      -> read_labels  -> Labels
    calib/<idx>.txt       "P2: ...", "R0_rect: ...", "Tr_velo_to_cam: ..." rows     -> read_calib   -> Calib
    labels + calib        boxes (n, 7) = (x, y, z, w, l, h, yaw) in the LIDAR frame -> boxes_in_lidar_frame
    points + calib        the part of the cloud the left colour camera sees        -> crop_to_camera_view

Values are computed with the reference's numpy expressions in the reference's dtypes (float32 calibration, float32
projection, float64 box conversion), so the golden test compares exactly.
"""
import os
from collections import namedtuple

import numpy as np

CLASS_INDEX = {"Car": 0, "Van": 0, "Pedestrian": 1, "Person_sitting": 1, "Cyclist": 2}  # anything else: -1 (ignored)
IMAGE_WH = (1224, 370)  # the reference's fixed (approximate) image size

Calib = namedtuple("Calib", ["V2C", "C2V", "R0", "P2", "WH"])  # field names of the reference's tuple: pickles stay readable
Labels = namedtuple("Labels", ["names", "class_idx", "truncation", "occlusion", "alpha", "box2d", "hwl", "location", "ry",
                               "score", "level"])


def read_points(path):
    """(N, 4) float32 view of a velodyne .bin file."""
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)


def _difficulty(box2d, truncation, occlusion):
    """KITTI difficulty 1 (easy) / 2 (moderate) / 3 (hard) / 4 (none of them), from the 2-D box height, truncation, occlusion."""
    height = box2d[:, 3] - box2d[:, 1] + 1
    level = np.full(len(height), 4, np.int64)
    for lv, (h_min, t_max, o_max) in ((3, (25, 0.5, 2)), (2, (25, 0.3, 1)), (1, (40, 0.15, 0))):
        level[(height >= h_min) & (truncation <= t_max) & (occlusion <= o_max)] = lv
    return level


def read_labels(path):
    """All objects of a label file as column arrays.  `location` is the box CENTRE in rectified-camera coordinates: the file
    stores the bottom centre, y points down, so the centre is y - h / 2 (the reference's Object3d.t)."""
    rows = [ln.rstrip().split(" ") for ln in open(path) if ln.strip()]
    names = [r[0] for r in rows]
    num = np.array([[float(v) for v in r[1:15]] for r in rows], np.float64).reshape(len(rows), 14)
    h, w, l = num[:, 7], num[:, 8], num[:, 9]
    location = np.stack((num[:, 10], num[:, 11] - h / 2, num[:, 12]), 1)
    score = np.array([float(r[15]) if len(r) == 16 else -1.0 for r in rows], np.float64)
    box2d = num[:, 3:7]
    occlusion = num[:, 1].astype(np.int64)
    return Labels(names=names, class_idx=np.array([CLASS_INDEX.get(n, -1) for n in names], np.int64), truncation=num[:, 0],
                  occlusion=occlusion, alpha=num[:, 2], box2d=box2d, hwl=np.stack((h, w, l), 1), location=location,
                  ry=num[:, 13], score=score, level=_difficulty(box2d, num[:, 0], occlusion))


def read_calib(path):
    """Calib(V2C (3,4), C2V (3,4), R0 (3,3), P2 (3,4), WH): velodyne -> reference camera, its inverse, reference ->
    rectified camera, rectified camera -> image 2; float32 like the reference's arrays."""
    rows = {}
    for ln in open(path):
        key, _, vals = ln.partition(":")
        if vals.strip():
            rows[key.strip()] = np.array(vals.split(), dtype=np.float32)
    v2c = rows["Tr_velo_to_cam"].reshape(3, 4)
    c2v = np.zeros_like(v2c)  # inverse of the rigid transform [R | t]: [R^T | -R^T t]
    c2v[:, :3] = v2c[:, :3].T
    c2v[:, 3] = np.dot(-v2c[:, :3].T, v2c[:, 3])
    return Calib(V2C=v2c, C2V=c2v, R0=rows["R0_rect"].reshape(3, 3), P2=rows["P2"].reshape(3, 4), WH=np.r_[IMAGE_WH[0], IMAGE_WH[1]])


def boxes_in_lidar_frame(labels, calib):
    """(n, 7) float64 (x, y, z, w, l, h, yaw): centre moved rectified camera -> lidar (C2V [R0 t; 1]), yaw = -ry
    (kitti_dataset.py:74-79)."""
    out = np.zeros((len(labels.names), 7), np.float64)
    for i in range(len(labels.names)):
        out[i, :3] = calib.C2V @ np.r_[calib.R0 @ tuple(labels.location[i]), 1]
    out[:, 3], out[:, 4], out[:, 5] = labels.hwl[:, 1], labels.hwl[:, 2], labels.hwl[:, 0]
    out[:, 6] = -labels.ry
    return out


def crop_to_camera_view(calib, points):
    """Points in front of the sensor whose projection into image 2 falls inside [0, W] x [0, H] (kitti_utils.py:49-58)."""
    keep = points[:, 0] > 0
    xyz = points[keep, :3]
    one = np.ones_like(xyz[:, 0:1])
    cam = (calib.R0 @ calib.V2C) @ np.c_[xyz, one].T
    img = calib.P2 @ np.r_[cam, one.T]
    uv = (img / img[2:3])[:2].T
    keep[keep] &= ((uv >= 0) & (uv <= calib.WH)).all(1)
    return points[keep]


def load_frame(root, idx, reduced=True):
    """One annotated frame as the detector's item fields: points (N, 4) float32, boxes (n, 7), class_idx (n,), calib."""
    name = f"{int(idx):06d}"
    calib = read_calib(os.path.join(root, "calib", name + ".txt"))
    labels = read_labels(os.path.join(root, "label_2", name + ".txt"))
    points = read_points(os.path.join(root, "velodyne_reduced" if reduced else "velodyne", name + ".bin"))
    return dict(points=points, boxes=boxes_in_lidar_frame(labels, calib), class_idx=labels.class_idx, calib=calib, idx=int(idx))
