"""Kernel launches and device time per section of one PV_RCNN.inference frame (eager): where the ~270 launches of the main stream come from.
usage (on the GPU box): python tools/pvrcnn_launch_census.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import ProfilerActivity, profile
from vision3d_amd import synth
from vision3d_amd.core import AnchorGenerator, Preprocessor
from vision3d_amd.core.config import second_car_cfg
from vision3d_amd.detector import PV_RCNN
from vision3d_amd.ops import batched_nms_rotated

cfg = second_car_cfg()
torch.manual_seed(0)
model = PV_RCNN(cfg).cuda().eval()
anchors = AnchorGenerator(cfg).anchors.cuda()
pre = Preprocessor(cfg, seed=0)
cloud = torch.from_numpy(synth.make_cloud(0, 16384)).cuda()


def census(name, fn):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    ev = [e for e in prof.key_averages()]
    n = sum(e.count for e in ev)
    t = sum(e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total for e in ev)
    top = sorted(ev, key=lambda e: -e.count)[:4]
    print(f"{name:34s} {n:4d} launches {t:9.0f} us   " + "; ".join(f"{e.key[:38]} x{e.count}" for e in top))
    return out


with torch.no_grad():
    for _ in range(3):
        model.inference(pre(dict(points=[cloud], anchors=anchors)))
    item = census("preprocess (voxelizer)", lambda: pre(dict(points=[cloud], anchors=anchors)))
    item = census("stage 1: proposal()", lambda: model.proposal(item))
    feats = census("point_feature_extract (VSA + BEV)", lambda: model.point_feature_extract(item, item["_cnn_features"], item["_bev_map"]))
    boxes, scores, class_idx = census("stage1_proposals (top-k, decode)", lambda: model.stage1_proposals(item))
    pooled = census("roi_grid_pool", lambda: model.roi_grid_pool(boxes, item["keypoints"], feats, None))
    deltas, conf = census("refinement_layer", lambda: model.refinement_layer(item["points"], pooled, boxes))
    refined = census("apply_refinements (decode)", lambda: model.refinement_layer.apply_refinements(deltas, boxes))

    def tail():
        b, n = refined.shape[:2]
        sc = conf.sigmoid().reshape(-1)
        bx = refined.reshape(-1, refined.shape[-1])
        bidx = torch.arange(b, device=bx.device).repeat_interleave(n)
        cidx = class_idx.repeat(b)
        keep = batched_nms_rotated(bx[:, [0, 1, 3, 4, 6]].contiguous(), sc, cidx + cfg.NUM_CLASSES * bidx, 0.01)
        bx, bidx, cidx, sc = (x[keep] for x in (bx, bidx, cidx, sc))
        mask = model.proposal_layer._above_score_thresh(sc, cidx)
        return [x[mask] for x in (bx, bidx, cidx, sc)]
    census("NMS + score cut", tail)
