"""Anchor grids captured from the REFERENCE generator (run in the build container only).

    python tests/golden/make_golden_anchors.py

Imports /root/reference/vision3d/core/anchor_generator.py by path (it needs nothing but torch) and stores, for the car-only
configuration (configs/second/car.yaml) and for the default 3-class configuration (core/config.py:14-50), the shape, two
checksums, the per-class centre z and 512 probed rows of `AnchorGenerator(cfg).anchors`.  The 3-class case pins a quirk:
make_anchor_centers writes the per-class z through an EXPANDED view (anchor_generator.py:56-58), so every class ends up with
the LAST class's center_z.  Only data is written (tests/golden/anchors.npz).
"""
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from vision3d_amd.core.config import _defaults, second_car_cfg  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_anchor_generator", "/root/reference/vision3d/core/anchor_generator.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
rng = np.random.default_rng(0)
perm = _defaults()
perm.ANCHORS = [perm.ANCHORS[i] for i in (2, 0, 1)]  # last class = Pedestrian: the "last class wins" reading, not "smallest z"
for tag, cfg in (("car", second_car_cfg()), ("three", _defaults()), ("three_perm", perm)):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = ref.AnchorGenerator(cfg).anchors
    flat = a.reshape(-1, 7)
    idx = np.sort(rng.choice(flat.shape[0], 512, replace=False))
    out[tag + "_shape"] = np.array(a.shape)
    out[tag + "_sum"] = np.array([a.double().sum().item(), a.double().abs().sum().item()])
    out[tag + "_class_z"] = a[:, 0, 0, 0, 2].numpy()
    out[tag + "_probe_idx"], out[tag + "_probe"] = idx, flat[idx].numpy()
np.savez_compressed(os.path.join(HERE, "anchors.npz"), **out)
print({k: v.shape for k, v in out.items()}, out["three_class_z"], out["three_perm_class_z"])
