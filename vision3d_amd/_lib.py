"""ctypes binding of libvision3d_hip.so (the C ABI declared in include/vision3d_hip.h).

This is the only place the package touches native code.  There is NO fallback: if the shared library
is missing or a tensor is not on the GPU, the call raises -- the product path never routes through a
CPU implementation (oracle/ is test infrastructure and is not importable from here).
"""
import ctypes as C
import contextlib
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# V3D_HIP_LIB: another build of the SAME library (A/B measurements of kernel changes on one box); never a fallback
LIB_PATH = os.environ.get("V3D_HIP_LIB") or os.path.join(_HERE, "lib", "libvision3d_hip.so")

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/vision3d_hip.h one to one
_SIGNATURES = {
    "v3d_version": (C.c_char_p, []),
    "v3d_hip_runtime_version": (_i, []),
    "v3d_compiler_version": (C.c_char_p, []),
    "v3d_error_string": (C.c_char_p, [_i]),
    "v3d_box_iou_rotated": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "v3d_box_iou_rotated_3d": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "v3d_nms_rotated_workspace": (_sz, [_i]),
    "v3d_nms_rotated": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "v3d_proposals_workspace": (_sz, [_i, _i, _i]),
    "v3d_proposals": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_proposals_flag": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_proposals_topk": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "v3d_refine_nms_workspace": (_sz, [_i, _i, _i]),
    "v3d_refine_nms": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_sparse_bn_workspace": (_sz, [_i, _i]),
    "v3d_sparse_bn_relu_fwd": (_i, [_vp, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _sz, _vp]),
    "v3d_sparse_bn_relu_bwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_assign_targets_workspace": (_sz, [_i, _i, _i]),
    "v3d_assign_targets": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_points_in_boxes": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "v3d_augment_work_bytes": (_sz, [_i, _i, _i]),
    "v3d_augment_frame": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, C.c_double, C.c_double, C.c_double, C.c_double,
                               _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_voxelize_workspace": (_sz, [_i]),
    "v3d_voxelize": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_rulebook_workspace": (_sz, [_i, _i, _i]),
    "v3d_rulebook_subm": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_rulebook_sparse": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "v3d_sparse_conv_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "v3d_sparse_conv_weight_image_bytes": (_sz, [_i, _i, _i]),
    "v3d_sparse_conv_pack_weights": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "v3d_sparse_conv_fwd_packed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v3d_sparse_rows_split": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "v3d_sparse_brick_table_bytes": (_sz, [_i, _i, _vp, _vp, _vp, _vp]),
    "v3d_sparse_brick_plan": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "v3d_sparse_conv_fwd_brick": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "v3d_act_scale_from_rows": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "v3d_rulebook_transpose": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "v3d_sparse_conv_bwd_weight_workspace": (_sz, [_i, _i, _i]),
    "v3d_sparse_conv_bwd_weight": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "v3d_densify": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "v3d_fps_workspace": (_sz, [_i, _i]),
    "v3d_furthest_point_sample": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "v3d_gather_points": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "v3d_ball_query": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _f, _i, _vp, _vp]),
    "v3d_ball_query_grid_workspace": (_sz, [_i, _i]),
    "v3d_ball_query_grid": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _f, _i, _vp, _vp, _sz, _vp]),
    "v3d_ball_query_grid_build": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "v3d_ball_query_grid_query_many": (_i, [_i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v3d_ball_query_grid_query": (_i, [_vp, _i, _i, _i, _f, _i, _vp, _f, _i, _vp, _vp, _sz, _vp]),
    "v3d_group_points": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "v3d_bev_bilinear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "v3d_sa_mlp_layer": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "v3d_sa_mlp_pair": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "v3d_voxel_centers": (_i, [_vp, _i, _f, _f, _f, _f, _f, _f, _vp, _vp]),
    "v3d_sa_mlp_pair2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                              _vp, _vp, _i, _i, _vp]),
    "v3d_roi_grid_points": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "v3d_linear_rows_many": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v3d_linear_rows": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "v3d_bev_gather_keypoints": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _i, _vp]),
    "v3d_backbone_create": (_i, [_vp, _vp, _vp]),
    "v3d_backbone_destroy": (None, [_vp]),
    "v3d_backbone_arena_bytes": (_sz, [_vp]),
    "v3d_backbone_tune": (_i, [_vp]),
    "v3d_backbone_num_layers": (_i, [_vp]),
    "v3d_backbone_set_layer": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "v3d_backbone_layer_output": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v3d_backbone_occupancy": (_vp, [_vp]),
    "v3d_backbone_overflow_flags": (_vp, [_vp]),
    "v3d_backbone_forward": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "v3d_backbone_forward_reuse": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "v3d_backbone_forward_voxels": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "v3d_backbone_train_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "v3d_backbone_train_backward": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "v3d_backbone_train_arena_bytes": (_sz, [_vp]),
    "v3d_backbone_tune_from_voxels": (_i, [_vp, _vp, _i, _i, _vp]),
    "v3d_bev_occupancy_words": (_sz, [_i, _i, _i]),
    "v3d_bev_occupancy_bits": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "v3d_backbone_bev_occupancy": (_vp, [_vp]),
    "v3d_backbone_bev_planes": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "v3d_backbone_set_throughput_mode": (_i, [_vp, _i]),
    "v3d_conv2d_bg_tiles": (_i, [_i, _i, _i]),
    "v3d_conv2d_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "v3d_conv2d_nhwc_split": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "v3d_conv2d_1x1_head_fused": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "v3d_nchw_to_split_nhwc": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "v3d_backbone_set_precision": (_i, [_vp, _i]),
    "v3d_backbone_precision": (_i, [_vp]),
    "v3d_backbone_set_presplit": (_i, [_vp, _i]),
    "v3d_backbone_act_scales": (_vp, [_vp]),
    "v3d_backbone_set_calibrating": (_i, [_vp, _i]),
    "v3d_backbone_calibrate": (_i, [_vp, _i, _vp]),
    "v3d_proposal_loss_workspace": (_sz, []),
    "v3d_proposal_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "v3d_proposal_loss_scale": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "v3d_conv2d_weight_image_bytes": (_sz, [_i, _i, _i]),
    "v3d_densify_nhwc_split": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "v3d_split_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "v3d_dense_train_weight_image_bytes": (_sz, [_i]),
    "v3d_dense_train_pack_weights": (_i, [_vp, _i, _i, _vp, _vp]),
    "v3d_dense_train_conv_tiles": (_i, [_i, _i, _i]),
    "v3d_dense_train_conv": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "v3d_dense_train_bn_finalize": (_i, [_vp, _i, C.c_longlong, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v3d_dense_train_bn_relu_apply": (_i, [_vp, C.c_longlong, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "v3d_dense_train_bn_bwd_workspace": (_sz, []),
    "v3d_dense_train_bn_relu_bwd": (_i, [_vp, _vp, C.c_longlong, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_dense_train_wgrad_workspace": (_sz, [_i]),
    "v3d_dense_train_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "v3d_dense_train_head_workspace": (_sz, [_i]),
    "v3d_dense_train_head_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "v3d_dense_train_head_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "v3d_dense_train_arena_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "v3d_dense_train_arena_init": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "v3d_dense_train_forward": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "v3d_dense_train_arena_bytes_split": (_sz, [_i, _i, _i, _i, _i]),
    "v3d_dense_train_forward_split": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "v3d_dense_train_backward_split": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "v3d_dense_train_backward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
}


class LayerDesc(C.Structure):
    _fields_ = [("subm", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("ksize", C.c_int32 * 3),
                ("stride", C.c_int32 * 3), ("padding", C.c_int32 * 3), ("key", C.c_int32), ("relu", C.c_int32)]


class BackboneConfig(C.Structure):
    _fields_ = [("voxel_size", C.c_float * 3), ("bounds", C.c_float * 6), ("max_pts", C.c_int32),
                ("max_voxels", C.c_int32), ("point_channels", C.c_int32), ("grid_shape", C.c_int32 * 3),
                ("max_batch", C.c_int32), ("max_points", C.c_int32), ("n_layers", C.c_int32), ("growth", C.c_float),
                ("conv_algo", C.c_int32)]



class Conv2dPrec(C.Structure):
    """v3d_conv2d_prec: arithmetic of a dense-head call + the device scale entries of its planes (f16s)."""
    _fields_ = [("prec", C.c_int32), ("in_entry", _vp), ("out_entry", _vp), ("range_flag", _vp), ("w_inv", _vp), ("w_inv2", _vp)]


PREC_BF16X3, PREC_F16S = 0, 1
PRECISIONS = {"bf16x3": PREC_BF16X3, "fp32": PREC_F16S, "f16s": PREC_F16S}


class TrainLayer(C.Structure):
    """v3d_train_layer: device pointers of one conv + BatchNorm group (parameters in, gradients out)."""
    _fields_ = [("weight", _vp), ("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp),
                ("num_batches_tracked", _vp), ("eps", C.c_float), ("momentum", C.c_float), ("grad_weight", _vp),
                ("grad_gamma", _vp), ("grad_beta", _vp)]


class DenseTrainLayer(C.Structure):
    """v3d_dense_train_layer: device pointers of one RPN conv + BatchNorm2d group (parameters in, gradients out)."""
    _fields_ = [("weight", _vp), ("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp),
                ("num_batches_tracked", _vp), ("eps", C.c_float), ("momentum", C.c_float), ("ksize", C.c_int32),
                ("grad_weight", _vp), ("grad_gamma", _vp), ("grad_beta", _vp)]


_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the native library; raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C vision3d_amd/csrc).  vision3d_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(code, what):
    if code != 0:
        msg = lib().v3d_error_string(int(code)).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The current HIP stream of the current device as an integer handle (tools/mb_host_call.py: 2.75 us through
    torch.cuda.current_stream(), 0.3 us through the raw accessor -- an eager PV-RCNN frame makes ~100 operator calls)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_NULL_GUARD = contextlib.nullcontext()


def device_guard(device):
    """`with device_guard(t.device):` -- torch.cuda.device(device) only when it is not the current device already (the usual case:
    0.4 instead of 1.3 us per operator call)."""
    index = device.index if isinstance(device, torch.device) else torch.device(device).index
    if index is None or index == torch.cuda.current_device():
        return _NULL_GUARD
    return torch.cuda.device(index)


def ptr(t):
    return 0 if t is None else t.data_ptr()


def require_gpu(name, *tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"{name}: tensors must live on the GPU (vision3d_amd has no CPU path); got {t.device}")


def as_f32(name, t):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def as_i32(name, t):
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name}: expected int32, got {t.dtype}")
    return t.contiguous()


def host_i32(values):
    """Small host-side int32 array argument (spatial shapes, kernel sizes, frame offsets)."""
    arr = (C.c_int32 * len(values))(*[int(v) for v in values])
    return arr


def host_f32(values):
    return (C.c_float * len(values))(*[float(v) for v in values])


_scale_scratch = {}


def scale_scratch(device):
    """The two zeroed uint32 words v3d_act_scale_from_rows works in, one pair per (device, stream): the kernel leaves them zero,
    so launches on one stream reuse them; two streams never share a pair."""
    if torch.cuda.is_current_stream_capturing():
        # a captured graph may replay beside other replays of graphs captured on this same stream: a pair of its own, zeroed by a
        # fill node of the graph itself
        return torch.zeros(2, dtype=torch.int32, device=device)
    index = torch.device(device).index
    key = (torch.cuda.current_device() if index is None else index, torch.cuda.current_stream(device).cuda_stream)
    buf = _scale_scratch.get(key)
    if buf is None:
        buf = _scale_scratch[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return buf


def workspace(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)
