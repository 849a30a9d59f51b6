"""Seeded synthetic lidar sweeps shaped like the clouds the reference consumes.

There is no KITTI data in this environment, so bench.py, the tests and smoke() use this ray-cast
generator (spec: SURVEY.md section 8(d)).  A sweep is what `dataset/kitti_dataset.py:135-140`
hands to the preprocessor: float32 (N, 4) = x, y, z, intensity, already cropped to GRID_BOUNDS.

Scene: sensor at the origin, ground plane z = -1.73 m, two walls parallel to the x axis and a set of
axis-aligned car-sized boxes; nearest hit per ray, gaussian range noise, half-open crop to the grid
bounds, shuffle, first `n_points`.
"""
import numpy as np

KITTI_BOUNDS = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)
WAYMO_BOUNDS = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
CAR_WLH = (1.6, 3.9, 1.56)


def _ray_scene(rng, az, el, n_cars, x_rng, y_rng, wall_rng):
    """Return (hit points (R,3), hit mask (R,), car boxes (n_cars,7) as x,y,z,w,l,h,yaw)."""
    a, e = np.meshgrid(az, el, indexing="ij")
    a, e = a.ravel(), e.ravel()
    d = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], 1)
    big = 1e9
    t = np.full(d.shape[0], big)
    # ground
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(d[:, 2] < 0, -1.73 / d[:, 2], big)
    t = np.minimum(t, tg)
    # walls y = +-yw, 4 m tall from the ground
    for sgn in (+1.0, -1.0):
        yw = sgn * rng.uniform(*wall_rng)
        with np.errstate(divide="ignore", invalid="ignore"):
            tw = np.where(d[:, 1] * sgn > 1e-9, yw / d[:, 1], big)
        zw = tw * d[:, 2]
        tw = np.where((zw >= -1.73) & (zw <= 2.27), tw, big)
        t = np.minimum(t, tw)
    # axis-aligned cars (random 90 degree swap of footprint)
    w, l, h = CAR_WLH
    boxes = np.zeros((n_cars, 7))
    for i in range(n_cars):
        cx, cy = rng.uniform(*x_rng), rng.uniform(*y_rng)
        swap = rng.random() < 0.5
        sx, sy = (w, l) if swap else (l, w)  # extent along x / y
        boxes[i] = (cx, cy, -0.95, w, l, h, np.pi / 2 if swap else 0.0)
        lo = np.array([cx - sx / 2, cy - sy / 2, -0.95 - h / 2])
        hi = np.array([cx + sx / 2, cy + sy / 2, -0.95 + h / 2])
        with np.errstate(divide="ignore", invalid="ignore"):
            t0, t1 = lo / d, hi / d
        tn = np.nanmax(np.minimum(t0, t1), 1)
        tf = np.nanmin(np.maximum(t0, t1), 1)
        tb = np.where((tn <= tf) & (tn > 0), tn, big)
        t = np.minimum(t, tb)
    hit = t < big
    t = t + rng.normal(0.0, 0.02, t.shape)
    return d * t[:, None], hit, boxes


def _morton_key(pts, bounds, vs=(0.05, 0.05, 0.1)):
    c = np.floor((pts[:, :3].astype(np.float32) - np.asarray(bounds[:3], np.float32)) / np.asarray(vs, np.float32)).astype(np.int64)
    key = np.zeros(pts.shape[0], np.uint64)
    for j in range(3):
        x = c[:, j].astype(np.uint64)
        for i in range(13):
            key |= ((x >> np.uint64(i)) & np.uint64(1)) << np.uint64(3 * i + j)
    return key


def make_cloud(seed=0, n_points=16384, bounds=KITTI_BOUNDS, fov_deg=45.0, az_steps=1000, n_beams=64,
               n_cars=12, return_boxes=False, order="shuffled"):
    """One synthetic sweep, float32 (n_points, 4).  Deterministic in `seed`.

    order="shuffled" (default; what every test and golden fixture uses): the returns in random order -- what the TRAINING
    dataset hands over (`kitti_dataset.py:154` shuffles the points).  order="scan": the SAME returns in firing order (azimuth
    step by azimuth step, 64 beams each) -- what the inference dataset hands over (`kitti_dataset.py:135-140`: `read_velo`,
    the .bin file in sensor order)."""
    rng = np.random.default_rng(seed)
    bounds = np.asarray(bounds, np.float64)
    x_rng = (max(bounds[0], -65.0) + 6.0, min(bounds[3], 71.0) - 5.4) if bounds[0] >= 0 else (bounds[0] + 8, bounds[3] - 8)
    y_rng = (-25.0, 25.0) if bounds[1] >= -40.0 else (bounds[1] + 8, bounds[4] - 8)
    el = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    steps = az_steps
    while True:
        sub = np.random.default_rng(rng.integers(1 << 31))
        az = np.deg2rad(np.linspace(-fov_deg, fov_deg, steps, endpoint=(fov_deg < 180.0)))
        pts, hit, boxes = _ray_scene(sub, az, el, n_cars, x_rng, y_rng, (10.0, 22.0))
        pts = pts[hit].astype(np.float32)
        lo, hi = bounds[:3].astype(np.float32), bounds[3:].astype(np.float32)
        keep = np.all((pts >= lo) & (pts < hi), 1)
        pts = pts[keep]
        if pts.shape[0] >= n_points:
            break
        steps *= 2  # denser azimuth sampling until the crop holds enough returns
    if order in ("scan", "morton"):  # the same subset as the shuffled sweep (same generator draws), restored to firing order
        sel = np.arange(pts.shape[0])
        sub.shuffle(sel)
        pts = pts[np.sort(sel[:n_points])]
        if order == "morton":  # (experiments: the returns sorted by the 3-D Morton key of their 0.05 x 0.05 x 0.1 m cell, stable)
            pts = pts[np.argsort(_morton_key(pts, bounds), kind="stable")]
    else:
        sub.shuffle(pts)
        pts = pts[:n_points]
    inten = sub.uniform(0.0, 1.0, (pts.shape[0], 1)).astype(np.float32)
    cloud = np.ascontiguousarray(np.concatenate([pts, inten], 1), dtype=np.float32)
    if return_boxes:
        return cloud, boxes.astype(np.float32)
    return cloud


def make_kitti_batch(batch_size=1, seed0=0, n_points=16384):
    return [make_cloud(seed0 + i, n_points) for i in range(batch_size)]


def make_waymo_cloud(seed=0, n_points=180000, order="shuffled"):
    """configs[4]: 360 degree sweep, 0.05 m voxels over +-75.2 m (HBM-bound stress case)."""
    return make_cloud(seed, n_points, WAYMO_BOUNDS, fov_deg=180.0, az_steps=3000, n_beams=64, n_cars=40, order=order)


def make_gt_boxes(seed=0, n_extra=15):
    """GT boxes (n,7) for the train configuration: the scene's cars + `n_extra` sampled ones
    (configs/second/car.yaml:18 AUG.NUM_SAMPLE_OBJECTS)."""
    _, boxes = make_cloud(seed, return_boxes=True)
    rng = np.random.default_rng(10_000 + seed)
    extra = np.zeros((n_extra, 7), np.float32)
    extra[:, 0] = rng.uniform(6, 65, n_extra)
    extra[:, 1] = rng.uniform(-25, 25, n_extra)
    extra[:, 2] = -0.95
    extra[:, 3:6] = np.asarray(CAR_WLH, np.float32) * rng.uniform(0.9, 1.1, (n_extra, 3))
    extra[:, 6] = rng.uniform(-np.pi, np.pi, n_extra)
    return np.concatenate([boxes, extra], 0).astype(np.float32)
