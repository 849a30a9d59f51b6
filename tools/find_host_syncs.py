"""Where does a PV-RCNN stage-2 step synchronise with the host?  Runs one step under torch.cuda.set_sync_debug_mode("warn") and
prints the source lines that triggered a synchronising HIP call (a pageable new_tensor in the BEV gatherer was the one that kept
frames on different streams from overlapping: 245 -> 760 frames/s once it was cached)."""
import sys, warnings, torch, numpy as np
sys.path.insert(0, ".")
from vision3d_amd import synth
from vision3d_amd.core import Preprocessor
from vision3d_amd.core.config import second_car_cfg
from vision3d_amd.detector import PV_RCNN
cfg = second_car_cfg(); torch.manual_seed(0)
model = PV_RCNN(cfg).cuda().eval()
with torch.no_grad():
    item = model.proposal(Preprocessor(cfg, seed=0)(dict(points=[synth.make_cloud(0, 16384)])))
    props = torch.from_numpy(np.stack([np.resize(synth.make_gt_boxes(0), (100, 7))])).cuda()
    def step():
        item["keypoints"] = model.sample_keypoints(item["points"])
        pf = model.point_feature_extract(item, item["_cnn_features"], item["_bev_map"])
        pooled = model.roi_grid_pool(props, item["keypoints"], pf)
        return model.refinement_layer(None, pooled, props)
    step(); torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        step()
    torch.cuda.set_sync_debug_mode("default")
    import traceback
    seen = set()
    for x in w:
        key = (x.filename, x.lineno)
        if key in seen: continue
        seen.add(key)
        print(x.filename.split("/")[-1], x.lineno, str(x.message)[:80])
