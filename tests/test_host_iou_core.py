"""CPU: the DEVICE geometry core (vision3d_amd/csrc/rotated_iou.h) compiled for the host with g++ must
reproduce the reference bit for bit (golden vectors + oracle/_ref when present).  This checks the
kernel's logic where no GPU exists; the GPU run of the same header is tests/test_gpu_iou_nms.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host_iou():
    so = os.path.join(HERE, "host", "_build", "libiou_host.so")
    src = os.path.join(HERE, "host", "iou_host_shim.cpp")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17", "-o", so, src])
    lib = C.CDLL(so)

    def run(b1, b2):
        """Both forms of the core -- work arrays as plain locals, and lane-interleaved in a 64-lane slab as the kernels keep
        them in LDS -- which must agree bit for bit; returns the plain one."""
        b1, b2 = np.ascontiguousarray(b1, np.float32), np.ascontiguousarray(b2, np.float32)
        out = np.empty((len(b1), len(b2)), np.float32)
        strided = np.empty_like(out)
        for fn, dst in ((lib.host_box_iou_rotated, out), (lib.host_box_iou_rotated_strided, strided)):
            fn(b1.ctypes.data_as(C.c_void_p), len(b1), b2.ctypes.data_as(C.c_void_p), len(b2), dst.ctypes.data_as(C.c_void_p))
        np.testing.assert_array_equal(out.view(np.uint32), strided.view(np.uint32))
        return out
    return run


@pytest.mark.parametrize("tag", ["kat", "deg", "rad", "far", "dense"])
def test_device_core_on_host_matches_golden(host_iou, golden_iou, tag):
    np.testing.assert_array_equal(host_iou(golden_iou[f"iou_{tag}_b1"], golden_iou[f"iou_{tag}_b2"]), golden_iou[f"iou_{tag}"])


def test_device_core_on_host_degenerates_and_oracle(host_iou, golden_iou, oracle):
    b = golden_iou["iou_degen_b"]
    np.testing.assert_array_equal(host_iou(b, b), golden_iou["iou_degen"])
    rng = np.random.default_rng(11)
    b1 = np.concatenate([rng.uniform(-3, 3, (400, 2)), rng.uniform(0.2, 5, (400, 2)), rng.uniform(-180, 180, (400, 1))], 1).astype(np.float32)
    b2 = b1.copy()
    b2[:, :2] += rng.normal(0, 1e-4, (400, 2)).astype(np.float32)
    np.testing.assert_array_equal(host_iou(b1, b2), oracle.box_iou_rotated(b1, b2))
    np.testing.assert_array_equal(host_iou(b1, b1), oracle.box_iou_rotated(b1, b1))
