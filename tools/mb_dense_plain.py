import sys, torch
sys.path.insert(0, ".")
from vision3d_amd.runtime import conv2d_split, pack_conv_weight, to_split_nhwc
torch.manual_seed(0)
x = torch.randn(1, 128, 200, 176, device="cuda")
w = torch.randn(128, 128, 3, 3, device="cuda") / 34
hi, lo = to_split_nhwc(x)
img = pack_conv_weight(w)
def f(): conv2d_split(hi, lo, img, None, True, 128, 128, 3)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): f()
e1.record(); torch.cuda.synchronize()
print(f"{10 * e0.elapsed_time(e1):.1f} us per conv")
