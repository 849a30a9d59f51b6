"""Rotated NMS timing on proposal-stage-like boxes (N = 1000 / 4096, car-sized boxes scattered over the KITTI BEV range)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vision3d_amd.ops import nms_rotated
from vision3d_amd import _lib as L
torch.manual_seed(0)
for n in (100, 1000, 4096):
    xy = torch.rand(n, 2, device="cuda") * torch.tensor([70.4, 80.0], device="cuda") + torch.tensor([0.0, -40.0], device="cuda")
    wl = torch.tensor([1.6, 3.9], device="cuda") * (0.8 + 0.4 * torch.rand(n, 2, device="cuda"))
    yaw = (torch.rand(n, 1, device="cuda") - 0.5) * 6.28
    boxes = torch.cat([xy, wl, yaw], 1).contiguous()
    scores = torch.rand(n, device="cuda")
    for mode in (0, 1):
      L.lib().v3d_debug_set_nms_rows(mode)
      for _ in range(3): keep = nms_rotated(boxes, scores, 0.01)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(20): keep = nms_rotated(boxes, scores, 0.01)
      e1.record(); torch.cuda.synchronize()
      print(f"N={n:5d} mask kernel {'per row (compaction)' if mode else 'per (row, block)'}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per nms_rotated call (host-driven, incl. sort + the keep-count read), kept {keep.numel()}")
