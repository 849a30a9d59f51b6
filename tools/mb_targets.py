import sys, time, torch
sys.path.insert(0, '/root/repo')
from vision3d_amd import synth
from vision3d_amd.core import ProposalTargetAssigner
from vision3d_amd.core.config import second_car_cfg
cfg = second_car_cfg()
a = ProposalTargetAssigner(cfg)
gt = torch.from_numpy(synth.make_gt_boxes(0)).cuda()
item = dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long).cuda(), box_ignore=torch.zeros(len(gt), dtype=torch.bool).cuda())
for name, fn in (("fused", a.forward), ("torch", a.forward_torch)):
    for _ in range(5): fn(dict(item))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): fn(dict(item))
    torch.cuda.synchronize(); print(name, f"{(time.perf_counter()-t)/50*1e6:.1f} us per frame ({len(gt)} gt x 70400 anchors)")
