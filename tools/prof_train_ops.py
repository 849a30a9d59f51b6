"""Which Python lines of a train step launch the small torch kernels (fills, copies, conversions)?  torch.profiler with stacks over
bench.py --mode train; prints, per (op, innermost frame inside this repository), the calls per step and the device time."""
import collections, os, runpy, sys, torch
from torch.profiler import profile, ProfilerActivity
STEPS = 4
sys.argv = ["bench.py", "--mode", "train", "--steps", str(STEPS), "--warmup", "3"]
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    runpy.run_path("bench.py", run_name="__main__")
agg = collections.defaultdict(lambda: [0, 0.0])
WANT = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::_to_copy", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::mul_",
        "aten::zeros", "aten::clone", "aten::contiguous", "aten::index_put_", "aten::sum", "aten::where", "aten::masked_fill_")
for e in prof.events():
    if e.name not in WANT or e.device_time_total <= 0:
        continue
    frame = next((f for f in e.stack if "/root/repo" in f or "vision3d_amd" in f or "bench.py" in f), e.stack[0] if e.stack else "?")
    k = (e.name, frame.strip()[-110:])
    agg[k][0] += 1
    agg[k][1] += e.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = STEPS + 3
for (name, frame), (n, us) in rows[:45]:
    print(f"{name:16s} {n / tot:5.1f}/step {us / tot:7.1f} us/step  {frame}")
