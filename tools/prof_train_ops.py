import os, sys, torch
sys.argv=["bench.py","--mode","train","--steps","3","--warmup","3"]
from torch.profiler import profile, ProfilerActivity
import runpy
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    runpy.run_path("bench.py", run_name="__main__")
ka = prof.key_averages(group_by_input_shape=True)
rows=[e for e in ka if e.key in ("aten::fill_","aten::zero_","aten::zeros","aten::copy_","aten::add","aten::add_","aten::mul","aten::mul_","aten::cat","aten::_to_copy","aten::contiguous","aten::clone","aten::zeros_like","aten::index_put_")]
rows.sort(key=lambda e:-e.count)
for e in rows[:60]:
    print(f"{e.key:18s} n={e.count:4d} cuda_us={e.device_time_total:9.1f} shapes={str(e.input_shapes)[:110]}")
