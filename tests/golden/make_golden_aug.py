"""Golden vectors for the GT-sampling augmentation, captured from the REFERENCE itself
(vision3d/dataset/augmentation.py, run in the build container only; see make_golden.py for how the reference's CPU
extension and Python files are made importable).

    python tests/golden/make_golden_aug.py      ->  tests/golden/augmentation.npz

The reference draws from the global numpy RNG without seeding; here every case seeds it, so the device implementation --
which makes the same draws in the same order -- can be compared value for value.  The reference's `.cuda()` inside
filter_collisions is made the identity for the run (no GPU here): the IoU then runs on the reference's own CPU operator.
Only DATA is written."""
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

from vision3d_amd import synth  # noqa: E402
from vision3d_amd.core.config import _defaults, second_car_cfg  # noqa: E402


def make_database(rng, per_class):
    """Synthetic sample database in the reference's format: {class: [dict(points (n,4) demeaned in xy, box (7,)), ...]}."""
    db = {}
    for c, k in enumerate(per_class):
        items = []
        for _ in range(k):
            w, l, h = rng.normal(1.6, 0.1), rng.normal(3.9, 0.3), rng.normal(1.56, 0.1)
            yaw = rng.uniform(-np.pi, np.pi)
            n = int(rng.integers(9, 60))
            local = rng.uniform(-0.5, 0.5, (n, 3)) * [w, l, h]
            cs, sn = np.cos(yaw), np.sin(yaw)
            xy = local[:, :2] @ np.array([[cs, sn], [-sn, cs]])
            z = local[:, 2:3] - 1.0
            pts = np.concatenate([xy, z, rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
            box = np.array([0, 0, -1.0, w, l, h, yaw], np.float32)
            items.append(dict(points=pts, box=box))
        db[c] = items
    return db


def main():
    refC = G.build_ref_C()
    G.install_stubs(refC)
    # the dataset package: importable by path, tqdm optional
    pkg = types.ModuleType("vision3d.dataset")
    pkg.__path__ = [G.REF + "/vision3d/dataset"]
    sys.modules["vision3d.dataset"] = pkg
    sys.modules.setdefault("tqdm", types.ModuleType("tqdm")).tqdm = lambda x, **k: x
    ops = G.load_file("vision3d.ops.iou_nms")
    sys.modules["vision3d.ops"].box_iou_rotated = ops.box_iou_rotated
    aug = G.load_file("vision3d.dataset.augmentation")
    torch.Tensor.cuda = lambda self, *a, **k: self  # no GPU in this container: the reference's IoU runs on its CPU operator

    out = {}
    cases = [("car", second_car_cfg(), [40, 0, 0], 11), ("car_b", second_car_cfg(), [40, 0, 0], 12),
             ("three", _defaults(), [30, 25, 25], 13), ("nosample", second_car_cfg(), [5, 0, 0], 14)]
    for tag, cfg, per_class, seed in cases:
        rng = np.random.default_rng(seed)
        db = make_database(rng, per_class)
        cache = tempfile.mkdtemp(prefix="v3d_aug_")
        with open(os.path.join(cache, "database.pkl"), "wb") as f:
            pickle.dump(db, f)
        cfg = cfg.clone()
        cfg.merge_from_dict(dict(DATA=dict(CACHEDIR=cache)))
        if tag == "nosample":
            cfg.AUG.DATABASE_SAMPLE = False
        points = synth.make_cloud(seed, 4096)
        gt = synth.make_gt_boxes(seed)[:9].astype(np.float32)
        cls = np.zeros(len(gt), np.int64)
        np.random.seed(1000 + seed)
        chain = aug.ChainedAugmentation(cfg)
        p, b, c = chain(points.copy(), gt.copy(), cls.copy())
        out[f"{tag}_points"], out[f"{tag}_boxes"], out[f"{tag}_class_idx"] = points, gt, cls
        out[f"{tag}_seed"] = np.array(1000 + seed)
        out[f"{tag}_out_points"], out[f"{tag}_out_boxes"], out[f"{tag}_out_class_idx"] = np.asarray(p), np.asarray(b), np.asarray(c)
        for ci, items in db.items():
            out[f"{tag}_db{ci}_points"] = np.concatenate([it["points"] for it in items]) if items else np.zeros((0, 4), np.float32)
            out[f"{tag}_db{ci}_sizes"] = np.array([len(it["points"]) for it in items], np.int64)
            out[f"{tag}_db{ci}_boxes"] = np.stack([it["box"] for it in items]) if items else np.zeros((0, 7), np.float32)
        out[f"{tag}_cfg"] = np.array([cfg.NUM_CLASSES, int(cfg.AUG.DATABASE_SAMPLE)] + list(cfg.AUG.NUM_SAMPLE_OBJECTS))
        print(tag, "in", points.shape, gt.shape, "->", np.asarray(p).shape, np.asarray(b).shape, np.asarray(p).dtype, np.asarray(b).dtype)
    np.savez_compressed(os.path.join(HERE, "augmentation.npz"), **out)


if __name__ == "__main__":
    main()
