"""Hyper-parameter tree for the hot path.

Mirrors the attribute surface of the reference's yacs global (`vision3d/core/config.py:4-110`:
`cfg.VOXEL_SIZE`, `cfg.PROPOSAL.TOPK`, `cfg.ANCHORS[i]['wlh']`, `cfg.merge_from_file(path)` ...)
without depending on yacs, which is not installed here.  Only the keys the hot path reads are
defaulted; unknown keys in a YAML override are accepted and stored.
"""
import copy
import math

import yaml


class Node(dict):
    """dict with attribute access; nested dicts become Nodes (lists of dicts stay plain dicts,
    because the reference indexes anchors as `anchor['wlh']`)."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = Node(v) if isinstance(v, dict) and not isinstance(v, Node) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return Node({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_dict(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), Node):
                self[k].merge_from_dict(v)
            else:
                self[k] = Node(v) if isinstance(v, dict) else v
        if "ANCHORS" in other and "NUM_CLASSES" not in other:
            self["NUM_CLASSES"] = len(self["ANCHORS"])
        return self

    def merge_from_file(self, path):
        with open(path) as f:
            return self.merge_from_dict(yaml.safe_load(f) or {})


def _defaults():
    half_pi = math.pi / 2
    anchors = [
        dict(names=["Car", "Van"], wlh=[1.6, 3.9, 1.56], yaw=[0, half_pi], iou_thresh=[0.45, 0.60],
             score_thresh=0.3, center_z=-1.0),
        dict(names=["Pedestrian", "Person_sitting"], wlh=[0.6, 0.8, 1.73], yaw=[0, half_pi],
             iou_thresh=[0.20, 0.35], score_thresh=0.3, center_z=-0.6),
        dict(names=["Cyclist"], wlh=[0.6, 1.76, 1.73], yaw=[0, half_pi], iou_thresh=[0.20, 0.35],
             score_thresh=0.3, center_z=-0.6),
    ]
    return Node(dict(
        C_IN=4, NUM_KEYPOINTS=2048, STRIDES=[1, 2, 4, 8], SAMPLES_PN=[16, 32],
        MAX_VOXELS=20000, MAX_OCCUPANCY=5, VOXEL_SIZE=[0.05, 0.05, 0.1],
        GRID_BOUNDS=[0, -40, -3, 70.4, 40, 1],
        CNN="SpMiddleFHD",
        ANCHORS=anchors, NUM_PROPOSAL_SAMPLE=-1, ALLOW_LOW_QUALITY_MATCHES=False,
        NUM_CLASSES=len(anchors), NUM_YAW=2, BOX_DOF=7,
        PSA=dict(
            RADII=[[0.4, 0.8], [0.4, 0.8], [0.8, 1.2], [1.2, 2.4], [2.4, 4.8]],
            MLPS=[[[1, 8, 16], [1, 8, 16]], [[4, 8, 16], [4, 8, 16]], [[32, 32, 32], [32, 32, 32]],
                  [[64, 64, 64], [64, 64, 64]], [[64, 64, 64], [64, 64, 64]]],
        ),
        GRIDPOOL=dict(NUM_GRIDPOINTS=16, RADII_PN=[0.8, 1.6], MLPS_PN=[[512, 192, 96], [512, 192, 96]],
                      MLPS_REDUCTION=[16 * 192, 256, 256]),
        PROPOSAL=dict(C_IN=128, TOPK=100),
        REFINEMENT=dict(MLPS=[256, 128]),
        TRAIN=dict(LR=1e-3, LAMBDA=1.0, EPOCHS=80, BATCH_SIZE=6, REFINEMENT_NUM_NEGATIVES=128),
        AUG=dict(GLOBAL_SCALE=[0.95, 1.05], GLOBAL_ROTATION=[-math.pi / 4, math.pi / 4], FLIP_HORIZONTAL=True,
                 DATABASE_SAMPLE=True, NUM_SAMPLE_OBJECTS=[15, 10, 10], MIN_NUM_SAMPLE_PTS=8),
    ))


# configs/second/car.yaml:1-18 restated as data: the single shipped override (car-only SECOND).
SECOND_CAR = dict(
    MAX_OCCUPANCY=5, MAX_VOXELS=20000, GRID_BOUNDS=[0, -40.0, -3, 70.4, 40.0, 1],
    ANCHORS=[dict(names=["Car", "Van"], wlh=[1.6, 3.9, 1.56], yaw=[0, 1.501], iou_thresh=[0.45, 0.60],
                  score_thresh=0.3, center_z=-1.0)],
    NUM_CLASSES=1,
    TRAIN=dict(BATCH_SIZE=4, LAMBDA=1.0, EPOCHS=60),
    AUG=dict(NUM_SAMPLE_OBJECTS=[15, 0, 0]),
)

# BASELINE.json configs[4]: Waymo-range sweep; MAX_VOXELS lifted.
WAYMO_RANGE = dict(GRID_BOUNDS=[-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], MAX_VOXELS=400000)


def second_car_cfg():
    return _defaults().merge_from_dict(copy.deepcopy(SECOND_CAR))


def waymo_range_cfg():
    return second_car_cfg().merge_from_dict(copy.deepcopy(WAYMO_RANGE))


cfg = _defaults()
