"""Run the two hot kernels a few times each (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vision3d_amd import synth, _lib as L
from vision3d_amd.core.config import second_car_cfg
from vision3d_amd.runtime import conv2d_split, pack_conv_weight, to_split_nhwc
from vision3d_amd.detector import Second
torch.manual_seed(0)
cfg = second_car_cfg()
model = Second(cfg).cuda().eval()
clouds = [torch.from_numpy(synth.make_cloud(0)).cuda()]
with torch.no_grad():
    for _ in range(5):
        model.head_maps_from_points(clouds)
torch.cuda.synchronize()
