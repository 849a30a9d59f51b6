"""CPU analysis behind csrc/brick.hip (round 6): how many DISTINCT input rows the 27 x R neighbours of R consecutive output rows are
(the LDS neighbourhood of a pass), and which share of (16-row tile, offset) products has any neighbour, for the submanifold layers
of the synthetic sweeps when ONLY the voxelizer's output is put in a spatial order (3-D Morton, or 8 x 8 / 16 x 16 voxel columns in
Morton order with an arbitrary order inside a column) and every later stage keeps the first-touch numbering of the strided
rulebooks -- i.e. what a plan in brick order would produce with one sort per frame.  Uses oracle/ (test infrastructure).
usage: python tools/brick_union_analysis.py [waymo|kitti]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from vision3d_amd import synth


def part(x, stride, bits=12):
    x = x.astype(np.uint64); r = np.zeros_like(x)
    for i in range(bits): r |= ((x >> np.uint64(i)) & np.uint64(1)) << np.uint64(stride * i)
    return r


def morton3(c): return part(c[:, 3], 3) | (part(c[:, 2], 3) << np.uint64(1)) | (part(c[:, 1], 3) << np.uint64(2))


def columns(b):
    return lambda c: part(c[:, 3] // b, 2) | (part(c[:, 2] // b, 2) << np.uint64(1))


def stages(cloud, bounds, keyfn, rng=None, vs=(0.05, 0.05, 0.1), max_vox=400000):
    v, co, num = orc.voxelize(cloud, vs, bounds, 5, max_vox)
    b = np.asarray(bounds)
    gx = np.round((b[3:] - b[:3]) / np.asarray(vs)).astype(int)
    shape = [int(gx[2]) + 1, int(gx[1]), int(gx[0])]
    coords = np.concatenate([np.zeros((co.shape[0], 1), np.int32), co], 1)
    if keyfn is not None:
        key = keyfn(coords)
        coords = coords[np.lexsort((rng.permutation(len(key)), key)) if rng is not None else np.argsort(key, kind="stable")]
    for s, (k, st, pd) in enumerate([(3, 2, 1), (3, 2, 1), (3, 2, [0, 1, 1]), ((3, 1, 1), (2, 1, 1), 0)]):
        yield f"subm{s}", orc.subm_rulebook(coords, shape, 3)
        coords, _, shape = orc.sparse_rulebook(coords, shape, k, st, pd)


def analyse(nbr, R, umax):
    n = nbr.shape[0]
    un = np.array([len(np.unique(nbr[s:s + R][nbr[s:s + R] >= 0])) for s in range(0, n, R)])
    m = nbr >= 0
    pad = (-n) % 16
    mm = np.concatenate([m, np.zeros((pad, m.shape[1]), bool)]) if pad else m
    return un, (un > umax).mean(), mm.reshape(-1, 16, m.shape[1]).any(1).mean(), m.mean()


which = sys.argv[1] if len(sys.argv) > 1 else "waymo"
cloud, bounds = (synth.make_waymo_cloud(0), synth.WAYMO_BOUNDS) if which == "waymo" else (synth.make_cloud(0), synth.KITTI_BOUNDS)
for name, kf, rng in (("first-touch order of the shuffled sweep", None, None), ("3-D Morton at stage 0", morton3, None),
                      ("8 x 8 columns (Morton), arbitrary inside, at stage 0", columns(8), np.random.default_rng(0)),
                      ("16 x 16 columns (Morton), arbitrary inside, at stage 0", columns(16), np.random.default_rng(0))):
    print(f"== {which}: {name}")
    for lname, nbr in stages(cloud, bounds, kf, rng):
        for R, umax in ((128, 240), (256, 480)):
            un, ov, tfrac, rfrac = analyse(nbr, R, umax)
            print(f"  {lname} rows {nbr.shape[0]:6d}  pass {R}: distinct rows / pass rows mean {un.mean() / R:.2f} p90 {np.percentile(un, 90) / R:.2f} p99 {np.percentile(un, 99) / R:.2f} "
                  f"max {un.max() / R:.2f}, passes beyond {umax} slots {ov:.3f} | (tile16, offset) with a neighbour {tfrac:.2f}, (row, offset) {rfrac:.2f}")
