"""Host time of one eager operator call and of its parts (the PV-RCNN end-to-end line is bound by ~100 such calls per frame)."""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from vision3d_amd import _lib as L
from vision3d_amd.pointnet2 import pointnet2_utils as PU


def t(fn, n=20000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt


dev = torch.device("cuda", 0)
a = torch.randn(64, 64, device=dev)
w = torch.randn(64, 16, device=dev)


def guard():
    with torch.cuda.device(dev):
        pass


null = contextlib.nullcontext()


def guard_cheap():
    with (null if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)):
        pass


print("with torch.cuda.device(dev)      %.2f us" % t(guard))
print("current_device() check + null    %.2f us" % t(guard_cheap))
print("L.stream_ptr()                   %.2f us" % t(L.stream_ptr))
print("torch.empty((64, 16))            %.2f us" % t(lambda: torch.empty((64, 16), dtype=torch.float32, device=dev)))
print("a.data_ptr()                     %.2f us" % t(a.data_ptr))
print("a.contiguous()                   %.2f us" % t(a.contiguous))
print("is_current_stream_capturing      %.2f us" % t(torch.cuda.is_current_stream_capturing))
out = torch.empty((64, 16), device=dev)
fn = L.lib().v3d_linear_rows
sp = L.stream_ptr()
print("ctypes call alone (linear_rows)  %.2f us" % t(lambda: fn(a.data_ptr(), 64, 64, 64, w.data_ptr(), None, 16, 0, out.data_ptr(), 16, 16, sp), 5000))
print("PU.linear_rows(a, w)             %.2f us" % t(lambda: PU.linear_rows(a, w), 5000))
print("torch.relu(a)  (reference)       %.2f us" % t(lambda: torch.relu(a), 5000))
