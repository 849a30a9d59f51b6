"""FPS 16 384 -> 2 048 (one frame and a batch of 8): time per call by HIP events, indices checked against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd import synth
from vision3d_amd.pointnet2.pointnet2_utils import furthest_point_sample
from oracle import oracle as O
for name, clouds in (("kitti x1", [synth.make_cloud(0)]), ("kitti x8", [synth.make_cloud(s) for s in range(8)]),
                     ("waymo crop 16k", [synth.make_waymo_cloud(0)[:16384]])):
    xyz = np.stack([c[:, :3] for c in clouds])
    d = torch.from_numpy(xyz).cuda()
    idx = furthest_point_sample(d, 2048)
    ok = np.array_equal(idx.cpu().numpy(), O.fps(xyz, 2048))
    for _ in range(3): furthest_point_sample(d, 2048)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): furthest_point_sample(d, 2048)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:16s} exact={ok}  {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per call")
