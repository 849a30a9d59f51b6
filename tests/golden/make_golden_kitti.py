"""Golden vectors for the KITTI file readers from the REFERENCE's own vision3d/dataset/kitti_utils.py
and the box conversion of kitti_dataset.py:74-79, run on synthetic files in the build container.

    python tests/golden/make_golden_kitti.py   ->  tests/golden/kitti.npz   (file CONTENTS as data + the reference's outputs)
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


def synth_files(rng, n_obj, n_pts):
    names = rng.choice(["Car", "Van", "Pedestrian", "Person_sitting", "Cyclist", "Truck", "Tram", "Misc", "DontCare"], n_obj)
    lines = []
    for i, nm in enumerate(names):
        trunc, occ, alpha = rng.uniform(0, 0.6), int(rng.integers(0, 4)), rng.uniform(-3.14, 3.14)
        x0, y0 = rng.uniform(0, 1100), rng.uniform(100, 300)
        bw, bh = rng.uniform(10, 200), rng.uniform(10, 120)
        h, w, l = rng.uniform(1.2, 2.0), rng.uniform(0.5, 2.0), rng.uniform(0.8, 5.0)
        x, y, z, ry = rng.uniform(-20, 20), rng.uniform(1.0, 2.0), rng.uniform(3, 60), rng.uniform(-3.14, 3.14)
        vals = [trunc, occ, alpha, x0, y0, x0 + bw, y0 + bh, h, w, l, x, y, z, ry]
        fields = [nm] + [f"{v:.2f}" if not isinstance(v, int) else str(v) for v in vals]
        if i % 3 == 0:
            fields.append(f"{rng.uniform(0, 1):.4f}")  # result files carry a score
        lines.append(" ".join(fields))
    label_txt = "\n".join(lines) + "\n"
    P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
    R0 = np.array([[0.9999239, 0.00983776, -0.007445048], [-0.009869795, 0.9999421, -0.004278459], [0.007402527, 0.004351614, 0.9999631]])
    V2C = np.array([[0.007533745, -0.9999714, -0.000616602, -0.004069766], [0.01480249, 0.0007280733, -0.9998902, -0.07631618],
                    [0.9998621, 0.00752379, 0.01480755, -0.2717806]])
    P2 = P2 + rng.normal(0, 1e-3, P2.shape)
    fmt = lambda a: " ".join(f"{v:.12e}" for v in np.asarray(a).ravel())
    calib_txt = (f"P0: {fmt(P2)}\nP1: {fmt(P2)}\nP2: {fmt(P2)}\nP3: {fmt(P2)}\nR0_rect: {fmt(R0)}\nTr_velo_to_cam: {fmt(V2C)}\n"
                 f"Tr_imu_to_velo: {fmt(V2C)}\n")
    pts = np.concatenate([rng.uniform(-10, 70, (n_pts, 1)), rng.uniform(-40, 40, (n_pts, 1)), rng.uniform(-3, 1, (n_pts, 1)),
                          rng.uniform(0, 1, (n_pts, 1))], 1).astype(np.float32)
    return label_txt, calib_txt, pts


def main():
    refC = G.build_ref_C()
    G.install_stubs(refC)
    pkg = types.ModuleType("vision3d.dataset")
    pkg.__path__ = [G.REF + "/vision3d/dataset"]
    sys.modules["vision3d.dataset"] = pkg
    ku = G.load_file("vision3d.dataset.kitti_utils")
    out = {}
    for case, (seed, n_obj, n_pts) in enumerate([(1, 7, 5000), (2, 1, 300), (3, 19, 20000)]):
        rng = np.random.default_rng(seed)
        label_txt, calib_txt, pts = synth_files(rng, n_obj, n_pts)
        d = tempfile.mkdtemp(prefix="v3d_kitti_")
        lp, cp, vp = (os.path.join(d, n) for n in ("l.txt", "c.txt", "v.bin"))
        open(lp, "w").write(label_txt)
        open(cp, "w").write(calib_txt)
        pts.tofile(vp)
        objs = ku.read_label(lp)
        calib = ku.read_calib(cp)
        k = f"c{case}_"
        out[k + "label_txt"], out[k + "calib_txt"], out[k + "points"] = np.array(label_txt), np.array(calib_txt), pts
        np.testing.assert_array_equal(ku.read_velo(vp), pts)
        for f in ("V2C", "C2V", "R0", "P2", "WH"):
            out[k + "calib_" + f] = np.asarray(getattr(calib, f))
        out[k + "class_idx"] = np.array([o.class_idx for o in objs])
        out[k + "level"] = np.array([o.level for o in objs])
        out[k + "t"] = np.array([o.t for o in objs], np.float64)
        out[k + "hwl"] = np.array([[o.h, o.w, o.l] for o in objs], np.float64)
        out[k + "misc"] = np.array([[o.truncation, o.occlusion, o.alpha, o.ry, o.score] for o in objs], np.float64)
        out[k + "box2d"] = np.stack([o.box2d for o in objs])
        # kitti_dataset.py:74-79 (AnnotationLoader._numpify_object), evaluated with the reference's expression
        out[k + "boxes"] = np.stack([np.r_[calib.C2V @ np.r_[calib.R0 @ o.t, 1], o.w, o.l, o.h, -o.ry] for o in objs])
        out[k + "fov_points"] = ku.filter_camera_fov(calib, pts)
        print(case, len(objs), pts.shape, "->", out[k + "fov_points"].shape)
    np.savez_compressed(os.path.join(HERE, "kitti.npz"), **out)


if __name__ == "__main__":
    main()
