"""Time per frame of the device-side GT-sampling + global augmentation (16 384-point cloud, 9 scene boxes, 15 pasted cars)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import synth
from vision3d_amd.core.config import second_car_cfg
from vision3d_amd.dataset import ChainedAugmentation
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "augmentation.npz"))
sizes = g["car_db0_sizes"]
db = {0: [dict(points=p, box=b) for p, b in zip(np.split(g["car_db0_points"], np.cumsum(sizes)[:-1]), g["car_db0_boxes"])], 1: [], 2: []}
cfg = second_car_cfg()
pts = torch.from_numpy(synth.make_cloud(0, 16384)).cuda()
boxes = torch.from_numpy(synth.make_gt_boxes(0)[:9]).cuda()
cls = torch.zeros(9, dtype=torch.int64, device="cuda")
from vision3d_amd.dataset import SampleDatabase
sdb = SampleDatabase(db, cfg.NUM_CLASSES)
for name, fused in (("fused (v3d_augment_frame)", True), ("class-by-class chain", False)):
    aug = ChainedAugmentation(cfg, database=sdb, fused=fused)
    np.random.seed(0)
    for _ in range(10): out = aug(pts, boxes, cls)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): out = aug(pts, boxes, cls)
    torch.cuda.synchronize()
    print(f"device augmentation, {name}: {(time.perf_counter() - t0) / 200 * 1e6:.0f} us per frame -> {out[0].shape[0]} points, {out[1].shape[0]} boxes")
