import sys, time, torch
sys.path.insert(0, '/root/repo')
from vision3d_amd import synth
from vision3d_amd.core import AnchorGenerator
from vision3d_amd.core.config import second_car_cfg
from vision3d_amd.detector import Second
cfg = second_car_cfg(); torch.manual_seed(0)
model = Second(cfg).cuda().eval()
anchors = AnchorGenerator(cfg).anchors.cuda()
clouds = [torch.from_numpy(synth.make_cloud(0, 16384)).cuda()]
for depth in (2, 3, 4):
    with torch.no_grad():
        pipe = model.pipelined_inference(anchors, [16384], depth)
        for _ in range(10): pipe(clouds)
        pipe.flush(); torch.cuda.synchronize()
        ts = tc = 0.0; n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            if len(pipe.pending) == len(pipe.slots):
                a = time.perf_counter(); pipe.collect(); tc += time.perf_counter() - a
            a = time.perf_counter(); pipe.submit(clouds); ts += time.perf_counter() - a
        pipe.flush(); torch.cuda.synchronize()
        tot = time.perf_counter() - t0
    print(f"depth {depth}: {tot/n*1e6:.0f} us/frame; submit {ts/n*1e6:.0f} us, collect {tc/n*1e6:.0f} us (incl. waiting)")
