"""CPU: the C-ABI library loads without a GPU and exports exactly what include/vision3d_hip.h declares."""
import ctypes
import os
import re

from vision3d_amd import _lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(REPO, "include", "vision3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(v3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 20
    handle = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/vision3d_hip.h but not exported"


def test_python_binding_covers_header():
    assert L.exported_symbols() == declared_functions()
    lib = L.lib()
    assert b"gfx950" in lib.v3d_version()
    assert lib.v3d_error_string(-1).startswith(b"invalid")
    assert lib.v3d_nms_rotated_workspace(4096) > 4096 * 64 * 8


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from vision3d_amd.ops import box_iou_rotated
    with pytest.raises(RuntimeError, match="GPU"):
        box_iou_rotated(torch.zeros(2, 5), torch.zeros(3, 5))
