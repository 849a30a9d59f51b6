"""Stand-in for the reference's pybind module `vision3d._C` (ops/csrc/vision.cpp:60-65): the same four
names, backed by the C ABI of libvision3d_hip.so."""
from . import _lib as L
from .ops.iou_nms import box_iou_rotated, nms_rotated  # noqa: F401


def get_compiler_version():
    return L.lib().v3d_compiler_version().decode()


def get_cuda_version():
    """The reference reports CUDART_VERSION (cuda_version.cu:6-8); here: the HIP runtime version."""
    v = L.lib().v3d_hip_runtime_version()
    return "not available" if v < 0 else f"HIP {v // 10000000}.{v // 100000 % 100}.{v % 100000}"
