"""Native runtime objects: the fused sparse-backbone plan (csrc/second_plan.hip).

`BackbonePlan(cnn, cfg)` mirrors one `SparseCNNBase` module (detector/sparse_cnn.py) inside
libvision3d_hip.so: voxelizer + every sparse layer (BatchNorm folded, ReLU fused) + .dense(), enqueued
by ONE C call per forward with no host synchronisation.  Parameters are re-uploaded automatically when
the module's tensors change (load_state_dict, optimizer step)."""
import ctypes as C
import os

import numpy as np
import torch
from torch import nn

from . import _lib as L
from .spconv.conv import _SparseConvBase
from .spconv.modules import fold_batchnorm


class RangeOverflow(RuntimeError):
    """f16s arithmetic: a tensor of the last frame exceeded the range its scale entry was calibrated for (the frame's results
    are not valid).  Recalibrate on that frame (BackbonePlan.recalibrate / DenseHeadPlan.recalibrate) and run it again --
    Second.inference and the captured-graph runners do."""


class RangeUnderflow(RangeOverflow):
    """f16s arithmetic, the other direction: a tensor of the last frame stayed 2^12 or more below the limit its scale entry was
    calibrated for (summary word 3, csrc/second_plan.hip plan_quiet_check_kernel) -- the frame's small entries no longer keep the
    22 bits of the fp32-class arithmetic.  Handled like RangeOverflow (which it subclasses, so every recalibrate-and-rerun path
    catches it): the entries are re-derived from this frame, downward, and the frame is run again."""


CALIB_HEADROOM_BITS = 5  # a later frame may exceed the calibration frame's maxima by 2^6 before the range flag is raised


def scale_entry_from_max(amax, headroom_bits, out=None):
    """(4,) float32 device entry {s, 1/s, 2^15 / s, max} for a tensor whose largest magnitude is the 0-dim tensor `amax`: s = the
    power of two that puts it into [2^(13 - h), 2^(14 - h)) -- csrc/spconv.hip v3d_pow2_scale, here as device-side torch ops (no
    host synchronisation)."""
    amax = amax.detach().to(torch.float32).reshape(())
    _, e = torch.frexp(amax)  # amax = m * 2^e, m in [0.5, 1)
    ok = torch.isfinite(amax) & (amax > 0)
    expo = torch.where(ok, (14 - int(headroom_bits)) - e, torch.zeros_like(e)).clamp(-125, 125)
    s = torch.ldexp(torch.ones((), dtype=torch.float32, device=amax.device), expo)
    entry = torch.stack((s, 1.0 / s, 32768.0 / s, amax))
    if out is not None:
        out.copy_(entry)
        return out
    return entry


class PlanCache(dict):
    """Per-module cache of native plans.  Plans own device arenas through a C handle: a deep copy of the module
    (copy.deepcopy(model), EMA / checkpoint helpers) starts with an empty cache instead of aliasing the handles."""

    def __deepcopy__(self, memo):
        return PlanCache()


def flatten_sparse_layers(module):
    """[(conv, bn|None, relu: bool)] in execution order from a tree of SparseSequential."""
    mods = []

    def walk(m):
        if isinstance(m, _SparseConvBase) or not isinstance(m, nn.Sequential):
            mods.append(m)
        else:
            for c in m.children():
                walk(c)
    walk(module)
    out, i = [], 0
    while i < len(mods):
        m = mods[i]
        if not isinstance(m, _SparseConvBase):
            raise NotImplementedError(f"BackbonePlan: unsupported module {type(m).__name__} outside a conv-BN-ReLU group")
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
        j = i + (2 if bn is not None else 1)
        relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
        out.append((m, bn, relu))
        i = j + (1 if relu else 0)
    return out


class BackbonePlan(object):

    def __init__(self, cnn, cfg, max_batch=1, max_points=None, growth=2.0, device=None, conv_algo=0, precision="bf16x3"):
        """precision: arithmetic of the packed layers in the inference entry points -- "fp32" (= "f16s": f16 hi / lo pieces under
        calibrated power-of-two scales, the reference's fp32 results up to summation noise) or "bf16x3" (bf16 pieces, 2^-17 per
        product, scale-free: the training plan and the fast mode).  csrc/spconv.hip "the split-precision product"."""
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.cfg = cfg
        self.layers = flatten_sparse_layers(cnn.blocks)
        self.grid_shape = [int(s) for s in cnn.grid_shape]
        self.max_batch = int(max_batch)
        self.max_points = int(max_points if max_points is not None else 16384 * max_batch)
        keys = {}
        descs = (L.LayerDesc * len(self.layers))()
        for d, (conv, bn, relu) in zip(descs, self.layers):
            d.subm, d.cin, d.cout = int(conv.subm), conv.in_channels, conv.out_channels
            d.ksize[:] = conv.kernel_size
            d.stride[:] = conv.stride
            d.padding[:] = conv.padding
            d.key = -1 if conv.indice_key is None else keys.setdefault(conv.indice_key, len(keys))
            d.relu = int(relu)
        c = L.BackboneConfig()
        c.voxel_size[:] = [float(v) for v in cfg.VOXEL_SIZE]
        c.bounds[:] = [float(v) for v in cfg.GRID_BOUNDS]
        c.max_pts, c.max_voxels, c.point_channels = cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS, cfg.C_IN
        c.grid_shape[:] = self.grid_shape
        c.max_batch, c.max_points, c.n_layers, c.growth = self.max_batch, self.max_points, len(self.layers), float(growth)
        c.conv_algo = int(conv_algo)
        self._handle = C.c_void_p()
        with L.device_guard(self.device):
            L.check(L.lib().v3d_backbone_create(C.byref(c), descs, C.byref(self._handle)), "backbone_create")
        self.precision = None
        self.set_precision(precision)
        if os.environ.get("V3D_PRESPLIT", "1") == "0":  # A/B measurements only (same results)
            self.set_presplit(False)
        self.out_channels = self.layers[-1][0].out_channels
        shape = list(self.grid_shape)
        for conv, _, _ in self.layers:
            if not conv.subm:
                shape = [(shape[j] + 2 * conv.padding[j] - conv.kernel_size[j]) // conv.stride[j] + 1 for j in range(3)]
        self.out_shape = shape
        self._stamp = None
        self._keep = []

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                L.lib().v3d_backbone_destroy(h)
            except Exception:
                pass
            self._handle = None

    # ---- arithmetic and its scale entries (f16s)
    def set_precision(self, precision):
        if precision not in L.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(L.PRECISIONS)}")
        if self.precision is not None and L.PRECISIONS[precision] == L.PRECISIONS[self.precision]:
            return
        L.check(L.lib().v3d_backbone_set_precision(self._handle, L.PRECISIONS[precision]), "backbone_set_precision")
        self.precision = precision
        self._stamp = None  # the weight images are packed per arithmetic
        self._calib = "need" if self.f16s else "off"

    @property
    def f16s(self):
        return L.PRECISIONS[self.precision] == L.PREC_F16S

    def set_presplit(self, on=True):
        """Packed layers also write their rows split for the next packed layer (default on; off = every layer splits the fp32 rows it
        gathers: A/B measurements).  Same results bit for bit."""
        L.check(L.lib().v3d_backbone_set_presplit(self._handle, int(bool(on))), "backbone_set_presplit")

    def act_scales(self):
        """(n_layers + 1, 4) float32 device view {s, 1/s, limit, max}: entry l = the rows layer l gathers, the last = the BEV map."""
        ptr = L.lib().v3d_backbone_act_scales(self._handle)
        return _view(ptr, (len(self.layers) + 1, 4), torch.float32, self.device)

    def bev_entry(self):
        """(4,) device entry of the split BEV planes (what DenseHeadPlan.forward takes as `in_entry`); None for bf16x3."""
        return self.act_scales()[len(self.layers)] if self.f16s else None

    def recalibrate(self):
        """f16s: the next eager forward re-derives the scale entries from its own frame (after a RangeOverflow)."""
        if self.f16s:
            self._calib = "need"

    def copy_calibration(self, other):
        """Take another plan's scale entries (the plans of a pipeline's slots share one calibration)."""
        if self.f16s and other.f16s and len(other.layers) == len(self.layers):
            self.act_scales().copy_(other.act_scales())
            self._calib = other._calib
            if other._calib == "done":
                L.check(L.lib().v3d_backbone_set_calibrating(self._handle, 0), "backbone_set_calibrating")
                self.calibration_generation = self.__dict__.get("calibration_generation", 0) + 1

    def _param_stamp(self):
        st = []
        for conv, bn, _ in self.layers:
            ts = [conv.weight, conv.bias] + ([bn.running_mean, bn.running_var, bn.weight, bn.bias] if bn is not None else [])
            st.append(tuple((t.data_ptr(), t._version) for t in ts if t is not None))
        return tuple(st)

    def sync_weights(self):
        """Upload folded parameters if any module tensor changed since the last upload."""
        stamp = self._param_stamp()
        if stamp == self._stamp:
            return
        if self.f16s and self._stamp is not None and self._calib == "done" and not torch.cuda.is_current_stream_capturing():
            # the scale entries were derived from the OLD weights' activations (load_state_dict, an optimizer step): derive them again
            # on the next eager frame -- the eager entry points that never read the range flag (forward, bev_from_points) would
            # otherwise run on stale entries; captured graphs are refreshed by their runner (detector/graph.py, weights_changed)
            self._calib = "need"
        lib = L.lib()
        keep = []
        with L.device_guard(self.device), torch.no_grad():
            for i, (conv, bn, _) in enumerate(self.layers):
                w = conv.weight.detach().to(self.device, torch.float32).reshape(-1, conv.in_channels, conv.out_channels).contiguous()
                scale = shift = None
                if bn is not None:
                    if bn.training:
                        raise RuntimeError("BackbonePlan runs eval-mode BatchNorm only (call model.eval())")
                    scale, shift = fold_batchnorm(bn)
                if conv.bias is not None:
                    b = conv.bias.detach()
                    shift = b if shift is None else shift + b * scale
                    scale = torch.ones_like(b) if scale is None else scale
                if scale is not None:
                    scale, shift = scale.to(self.device).contiguous(), shift.to(self.device).contiguous()
                keep += [w, scale, shift]
                L.check(lib.v3d_backbone_set_layer(self._handle, i, L.ptr(w), L.ptr(scale), L.ptr(shift), L.stream_ptr()),
                        "backbone_set_layer")
        self._keep = keep  # sources stay alive until the async copies have been consumed
        self._stamp = stamp

    def weights_changed(self):
        """True when a module tensor changed since the last upload (cheap: pointers and version counters)."""
        return self._stamp is not None and self._param_stamp() != self._stamp

    def forward(self, points, frame_offsets, out=None):
        """points (sum N, C) float32 cuda (frames concatenated), frame_offsets: host ints (B+1).
        Returns the BEV map (B, C_out * D, H, W); nothing is synchronised.  f16s: the scale entries follow the weights (sync_weights)
        and the first frame; a LATER frame outside the calibrated range raises the plan's summary word, which this entry point does not
        read -- callers that feed frames of very different magnitudes call `check_overflow()` (blocking) or use the inference paths."""
        self.sync_weights()
        pts = L.as_f32("backbone", points)
        b = len(frame_offsets) - 1
        d, h, w = self.out_shape
        if out is None:
            out = torch.empty((b, self.out_channels * d, h, w), dtype=torch.float32, device=pts.device)
        with L.device_guard(pts.device):
            L.check(L.lib().v3d_backbone_forward(self._handle, L.ptr(pts), pts.shape[0], L.host_i32(frame_offsets), b,
                                                 L.ptr(out), None, None, L.stream_ptr()), "backbone_forward")
        if self._maybe_tune():
            return self.forward(points, frame_offsets, out)
        return out

    def tune(self):
        """Refresh the kernel-choice size hints from the live row counts of the last forward (blocking; never during
        stream capture).  Capacities are upper bounds -- up to 30x the live count in late stages -- and the sparse
        kernels cross over at ~32 k live rows (csrc/spconv.hip)."""
        torch.cuda.synchronize(self.device)
        L.check(L.lib().v3d_backbone_tune(self._handle), "backbone_tune")
        self._tuned = True

    def _maybe_tune(self):
        """True while the forward that has just been enqueued must be repeated by the caller (eager calls only, never during
        stream capture):
          * once after the first forward: the kernels are now picked from the observed sparsity (the variants differ in the last
            bits), and -- that forward is synchronised anyway -- the capacities are checked against the observed frame;
          * f16s, first forward or after `recalibrate()`: one pass with every layer on the exact-fp32 kernel, from whose tensors
            the scale entries are derived (v3d_backbone_calibrate, device-side), then the frame itself."""
        if torch.cuda.is_current_stream_capturing():
            return False
        again = False
        if not self.__dict__.get("_tuned"):
            self.tune()
            if not self.__dict__.get("allow_overflow"):
                self.check_overflow(ignore_range=True)
            again = True
        if self._calib == "need":
            L.check(L.lib().v3d_backbone_set_calibrating(self._handle, 1), "backbone_set_calibrating")
            self._calib = "exact"
            return True
        if self._calib == "exact":
            L.check(L.lib().v3d_backbone_calibrate(self._handle, int(self.__dict__.get("calib_headroom", CALIB_HEADROOM_BITS)),
                                                   L.stream_ptr()), "backbone_calibrate")
            L.check(L.lib().v3d_backbone_set_calibrating(self._handle, 0), "backbone_set_calibrating")
            self._calib = "done"
            self.calibration_generation = self.__dict__.get("calibration_generation", 0) + 1
            return True
        return again

    def set_throughput_mode(self, on=True):
        """Kernels picked for frames that run BESIDE other frames on the GPU (v3d_backbone_set_throughput_mode): less CU-time per
        launch, slightly longer launches; bit-identical results.  PipelinedSecond sets it on the plans of its slots.  Takes effect
        at the next forward -- a captured graph keeps the kernels it was captured with."""
        L.check(L.lib().v3d_backbone_set_throughput_mode(self._handle, int(bool(on))), "backbone_set_throughput_mode")

    def own_planes(self, batch_size):
        """The plan's PERSISTENT split BEV planes as (B, H, W, C_out * D) int16 views (v3d_backbone_bev_planes): a forward into
        them clears only the pixels the previous frame wrote instead of filling both planes.  They alias plan memory: valid until
        the next forward of this plan, and nobody else may write them."""
        hi, lo = C.c_void_p(), C.c_void_p()
        L.check(L.lib().v3d_backbone_bev_planes(self._handle, C.byref(hi), C.byref(lo)), "backbone_bev_planes")
        d, h, w = self.out_shape
        full = (self.max_batch, h, w, self.out_channels * d)
        return self._tag(_view(hi.value, full, torch.int16, self.device)[:int(batch_size)],
                         _view(lo.value, full, torch.int16, self.device)[:int(batch_size)])

    def _tag(self, hi, lo):
        tag_planes(hi, self.bev_entry(), self.overflow_any() if self.f16s else None)
        return hi, lo

    def forward_split(self, points, frame_offsets, persistent=False):
        """Same as forward() but the BEV map comes out as the dense head's input format: two bf16 NHWC
        planes (B, H, W, C_out*D) hi/lo (stored as int16).  persistent: into the plan's own planes (`own_planes`) -- no per-frame
        fill of the map; what the captured graphs use (one plan per graph slot)."""
        self.sync_weights()
        pts = L.as_f32("backbone", points)
        b = len(frame_offsets) - 1
        d, h, w = self.out_shape
        hi, lo = self.own_planes(b) if persistent else self._tag(*split_planes_like(b, h, w, self.out_channels * d, pts.device))
        with L.device_guard(pts.device):
            L.check(L.lib().v3d_backbone_forward(self._handle, L.ptr(pts), pts.shape[0], L.host_i32(frame_offsets), b, 0,
                                                  L.ptr(hi), L.ptr(lo), L.stream_ptr()), "backbone_forward2")
        if self._maybe_tune():  # once: kernels are now picked by the observed sparsity
            return self.forward_split(points, frame_offsets, persistent)
        return hi, lo

    def forward_reuse_split(self, batch_size, device):
        """The convolutions of the frame forwarded last, on the rulebooks that call left in the plan (timing variant: no
        voxelizer, no rulebook build); same planes as forward_split returned."""
        d, h, w = self.out_shape
        hi, lo = self._tag(*split_planes_like(int(batch_size), h, w, self.out_channels * d, device))
        with L.device_guard(device):
            L.check(L.lib().v3d_backbone_forward_reuse(self._handle, int(batch_size), 0, L.ptr(hi), L.ptr(lo), L.stream_ptr()),
                    "backbone_forward_reuse")
        return hi, lo

    def forward_voxels_split(self, voxel_mean, coordinates, batch_size):
        """The plan fed with EXISTING voxels (`item['voxel_mean']` (M, C), `item['coordinates']` (M, 4) int32 of the
        Preprocessor) instead of raw points: split bf16 NHWC planes of the BEV map, as forward_split."""
        self.sync_weights()
        mean = L.as_f32("backbone", voxel_mean)
        coords = L.as_i32("backbone", coordinates)
        m = mean.shape[0]
        if coords.shape != (m, 4) or mean.shape[1] != self.cfg.C_IN:
            raise RuntimeError("backbone: voxel_mean (M, C_IN) / coordinates (M, 4) expected")
        d, h, w = self.out_shape
        hi, lo = self._tag(*split_planes_like(int(batch_size), h, w, self.out_channels * d, mean.device))
        with L.device_guard(mean.device):
            L.check(L.lib().v3d_backbone_forward_voxels(self._handle, L.ptr(mean), L.ptr(coords), m, int(batch_size), 0,
                                                        L.ptr(hi), L.ptr(lo), L.stream_ptr()), "backbone_forward_voxels")
        if self._maybe_tune():
            return self.forward_voxels_split(voxel_mean, coordinates, batch_size)
        return hi, lo

    def forward_voxels(self, voxel_mean, coordinates, batch_size, out=None):
        """The plan fed with existing voxels, BEV map as the reference's `.dense()` + fold returns it: (B, C_out * D, H, W) float32.
        The callers that also want the sparse levels (`layer_output`): PV_RCNN's stage 1."""
        self.sync_weights()
        mean = L.as_f32("backbone", voxel_mean)
        coords = L.as_i32("backbone", coordinates)
        m = mean.shape[0]
        if coords.shape != (m, 4) or mean.shape[1] != self.cfg.C_IN:
            raise RuntimeError("backbone: voxel_mean (M, C_IN) / coordinates (M, 4) expected")
        d, h, w = self.out_shape
        if out is None:
            out = torch.empty((int(batch_size), self.out_channels * d, h, w), dtype=torch.float32, device=mean.device)
        with L.device_guard(mean.device):
            L.check(L.lib().v3d_backbone_forward_voxels(self._handle, L.ptr(mean), L.ptr(coords), m, int(batch_size), L.ptr(out),
                                                        None, None, L.stream_ptr()), "backbone_forward_voxels")
        if self._maybe_tune():
            return self.forward_voxels(voxel_mean, coordinates, batch_size, out)
        return out

    def bev_occupancy(self, batch_size):
        """(B * H, ceil(W / 32)) int32 device view of the plan's BEV occupancy bitmap for the LAST forward_split /
        forward_voxels_split: one bit per BEV pixel, inverted (0 = occupied).  Reset by the plan's per-frame fill and written
        by its densify kernel (no extra launch); feeds the background-skipping dense head (DenseHeadPlan.forward(occ=...)).
        Aliases plan memory: valid until the next forward."""
        d, h, w = self.out_shape
        ptr = L.lib().v3d_backbone_bev_occupancy(self._handle)
        return _view(ptr, (int(batch_size) * h, (w + 31) // 32), torch.int32, self.device)

    # ---- training: the sparse half of a train step, one native call each way (csrc/second_plan.hip, "Training plan")
    def train_parameters(self):
        """[w0, gamma0, beta0, w1, ...]: the tensors the training plan differentiates, in the order of `train_forward`'s
        gradients."""
        out = []
        for conv, bn, _ in self.layers:
            out += [conv.weight, bn.weight, bn.bias]
        return out

    def train_supported(self):
        """Every layer conv (no bias) + BatchNorm1d in training mode with running statistics and a fixed momentum."""
        for conv, bn, _ in self.layers:
            if conv.bias is not None or bn is None or not bn.training or not bn.affine or bn.momentum is None:
                return False
            if (bn.running_mean is None) != (not bn.track_running_stats):
                return False
            c = conv.out_channels
            if c < 4 or c > 256 or (c & (c - 1)):
                return False
        return True

    def _train_io(self, grads=None):
        """Host array of v3d_train_layer for the CURRENT parameter tensors (+ gradient slices of one flat buffer)."""
        io = (L.TrainLayer * len(self.layers))()
        for i, (d, (conv, bn, _)) in enumerate(zip(io, self.layers)):
            for t in (conv.weight, bn.weight, bn.bias):
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                    raise RuntimeError("training plan: parameters must be contiguous float32 tensors on the plan's device")
            d.weight, d.gamma, d.beta = conv.weight.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr()
            if bn.track_running_stats and bn.running_mean is not None:
                d.running_mean, d.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                d.num_batches_tracked = bn.num_batches_tracked.data_ptr()
            d.eps, d.momentum = float(bn.eps), float(bn.momentum)
            if grads is not None:
                d.grad_weight, d.grad_gamma, d.grad_beta = (g.data_ptr() for g in grads[3 * i:3 * i + 3])
        return io

    def train_forward(self, voxel_mean, coordinates, batch_size, bf16_nhwc=False):
        """item voxels -> BEV map (B, C_out * D, H, W) [bf16_nhwc: bfloat16 in channels_last, what the autocast RPN consumes:
        no fp32 NCHW map, layout copy or cast of a 144 MB tensor per step] through conv + batch-statistics BatchNorm + ReLU layers; keeps what
        `train_backward` needs inside the plan.  No host synchronisation (except before the first call, which tunes the
        kernel choice from the row counts of a coordinate-only pass and checks the capacities)."""
        mean = L.as_f32("backbone", voxel_mean)
        coords = L.as_i32("backbone", coordinates)
        m, b = mean.shape[0], int(batch_size)
        if coords.shape != (m, 4) or mean.shape[1] != self.cfg.C_IN:
            raise RuntimeError("backbone: voxel_mean (M, C_IN) / coordinates (M, 4) expected")
        d, h, w = self.out_shape
        if bf16_nhwc:
            out = torch.empty((b, self.out_channels * d, h, w), dtype=torch.bfloat16, device=mean.device,
                              memory_format=torch.channels_last)
        else:
            out = torch.empty((b, self.out_channels * d, h, w), dtype=torch.float32, device=mean.device)
        io = self._train_io()
        with L.device_guard(mean.device):
            if not self.__dict__.get("_tuned") and not torch.cuda.is_current_stream_capturing():
                # first step: a coordinate-only pass picks the kernels from the row counts BEFORE the step runs (the step
                # itself is not repeatable: it updates the running statistics)
                L.check(L.lib().v3d_backbone_tune_from_voxels(self._handle, L.ptr(coords), m, b, L.stream_ptr()),
                        "backbone_tune_from_voxels")
                self._tuned = True
                if not self.__dict__.get("allow_overflow"):
                    self.check_overflow()
            if not torch.cuda.is_current_stream_capturing():
                self._check_deferred_overflow()  # the PREVIOUS step's capacity word, copied to pinned memory behind that step
            L.check(L.lib().v3d_backbone_train_forward(self._handle, L.ptr(mean), L.ptr(coords), m, b, io,
                                                       0 if bf16_nhwc else L.ptr(out), L.ptr(out) if bf16_nhwc else 0,
                                                       L.stream_ptr()), "backbone_train_forward")
            # the kernels updated the running statistics through raw pointers: bump the tensors' version counters so that every
            # cache keyed on (data_ptr, _version) -- folded BatchNorm of the inference plans -- sees the change even when no
            # optimizer step follows (statistics-only passes, skipped steps)
            stats = [t for _, bn, _ in self.layers if bn.track_running_stats and bn.running_mean is not None
                     for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked)]
            if stats:  # (a host-side counter bump: no kernel)
                torch._C._autograd._unsafe_set_version_counter(stats, [t._version + 1 for t in stats])
            if not torch.cuda.is_current_stream_capturing() and not self.__dict__.get("allow_overflow"):
                self._post_deferred_overflow()
        self.forward_generation = self.__dict__.get("forward_generation", 0) + 1
        return out

    # ---- capacity check of EVERY training step without a stall: the summary word of step s is copied to pinned host memory on
    # the step's stream and read when step s + 1 begins (by then the copy has long finished; the event query is the guard)
    def _post_deferred_overflow(self):
        st = self.__dict__.setdefault("_ovf_state", {})
        if "host" not in st:
            st["host"] = torch.zeros(1, dtype=torch.int32).pin_memory()
            st["event"] = torch.cuda.Event()
        st["host"].copy_(self.overflow_any(), non_blocking=True)
        st["event"].record()
        st["pending"] = True

    def check_deferred_overflow(self):
        """Raise if the LAST training forward dropped rows.  Its summary word was copied to pinned memory right behind that
        forward, so calling this after backward() and BEFORE optimizer.step() costs no stall (the copy finished while the
        backward was being enqueued) and keeps corrupted gradients out of the weights; the next train_forward calls it too, and
        a training script should call it once more after its last step (Second.check_train_overflow does both plans)."""
        self._check_deferred_overflow()

    def _check_deferred_overflow(self):
        st = self.__dict__.get("_ovf_state")
        if not st or not st.get("pending"):
            return
        st["event"].synchronize()  # recorded one whole step ago: returns at once
        st["pending"] = False
        if int(st["host"][0]) > 0:
            raise RuntimeError("sparse backbone (training plan): the last checked step exceeded an active-site capacity -- rows were "
                               "dropped, its BEV map, batch statistics and gradients are wrong; build the plan with a larger "
                               "`growth` (Middle.plan_growth) or set plan.allow_overflow to accept clamping")

    def train_backward(self, grad_bev, batch_size):
        """d(BEV) -> [dW0, dgamma0, dbeta0, dW1, ...] (views of one flat buffer), for the last `train_forward`."""
        if not self.__dict__.get("forward_generation"):
            raise RuntimeError("training plan: backward before any train_forward")
        # (the saved state is read, not consumed: a second backward of the SAME forward -- retain_graph -- is valid; a backward
        # of an EARLIER forward is caught by PlanTrainFunction through the generation counter)
        if grad_bev.dtype == torch.bfloat16:
            g, nhwc = grad_bev.contiguous(memory_format=torch.channels_last), True
        else:
            g, nhwc = L.as_f32("backbone", grad_bev), False
        params = self.train_parameters()
        flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=g.device)
        grads, off = [], 0
        for p in params:
            grads.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        io = self._train_io(grads)
        with L.device_guard(g.device):
            L.check(L.lib().v3d_backbone_train_backward(self._handle, 0 if nhwc else L.ptr(g), L.ptr(g) if nhwc else 0,
                                                        int(batch_size), io, L.stream_ptr()), "backbone_train_backward")
        return grads

    def layer_output(self, layer):
        """(features (cap, C) view, coords (cap, 4) view, n_rows device int32 (1,), shape) of the last forward;
        layer = -1 is the voxelizer output.  Views alias the plan's arena: valid until the next forward."""
        f, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap, ch = C.c_int(), C.c_int()
        shape = (C.c_int32 * 3)()
        L.check(L.lib().v3d_backbone_layer_output(self._handle, layer, C.byref(f), C.byref(c), C.byref(n), C.byref(cap),
                                                  C.byref(ch), shape), "backbone_layer_output")
        return _view(f.value, (cap.value, ch.value), torch.float32, self.device), \
            _view(c.value, (cap.value, 4), torch.int32, self.device), _view(n.value, (1,), torch.int32, self.device), list(shape)

    def overflow(self):
        """(n_layers + 1,) int32 device flags of the last forward: [l] = 1 where layer l hit its active-site capacity
        (rows were dropped), [n_layers] = 1 if any did (or an f16s tensor left its calibrated range: `overflow_any` reads 2)."""
        ptr = L.lib().v3d_backbone_overflow_flags(self._handle)
        return (_view(ptr, (len(self.layers) + 1,), torch.int32, self.device) > 0).to(torch.int32)

    def overflow_any(self):
        """(1,) int32 device view of the summary word (> 0 = some stage dropped rows in the last forward): what the
        inference paths read together with the proposal count (ProposalLayer.finalize_native)."""
        ptr = L.lib().v3d_backbone_overflow_flags(self._handle)
        return _view(ptr + 4 * len(self.layers), (1,), torch.int32, self.device)

    def check_overflow(self, ignore_range=False):
        """Blocking check for callers that only take the BEV map (no per-frame host read of their own).  The summary word is 1 when
        a capacity was hit, 2 when an f16s tensor left its calibrated range (RangeOverflow: recalibrate and re-run)."""
        word = int(self.overflow_any().item())
        if word == 3 and not ignore_range:
            raise RangeUnderflow("sparse backbone (f16s): a tensor stayed 2^12 below its calibrated range; recalibrate() and run the frame again")
        if word == 2 and not ignore_range:
            raise RangeOverflow("sparse backbone (f16s): a tensor exceeded its calibrated range; recalibrate() and run the frame again")
        if word == 1 or (word >= 2 and any(self.overflow()[:-1].tolist())):
            hit = [i for i, f in enumerate(self.overflow()[:-1].tolist()) if f]
            raise RuntimeError(f"sparse backbone: layers {hit} exceeded their active-site capacity (rows were dropped); "
                               "build the plan with a larger `growth`")


class PlanTrainFunction(torch.autograd.Function):
    """Autograd node for the whole sparse backbone: forward / backward are ONE native call each (BackbonePlan.train_*).
    The parameters are passed so that autograd routes their gradients; the plan reads them through the modules."""

    @staticmethod
    def forward(ctx, plan, voxel_mean, coordinates, batch_size, bf16_nhwc, *params):
        ctx.plan, ctx.batch_size = plan, int(batch_size)
        out = plan.train_forward(voxel_mean.detach(), coordinates, batch_size, bf16_nhwc)
        ctx.generation = plan.forward_generation  # the plan holds the saved state of ONE forward: see backward
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_bev):
        if ctx.generation != ctx.plan.forward_generation:
            raise RuntimeError("training plan: this backward belongs to forward #%d but the plan now holds the state of forward #%d "
                               "(two train-mode forwards through one plan before a backward: the saved activations and rulebooks "
                               "were overwritten).  Run one forward/backward pair at a time per plan, or set "
                               "Middle.native_train = False for this pattern." % (ctx.generation, ctx.plan.forward_generation))
        hook = ctx.plan.__dict__.get("pre_backward_hook")
        if hook is not None:  # everything downstream of the BEV map has its gradients by now (dist_util.TwoPhaseGradReducer)
            hook()
        grads = ctx.plan.train_backward(grad_bev, ctx.batch_size)
        return (None, None, None, None, None) + tuple(grads)


class _DevMem(object):
    """Minimal __cuda_array_interface__ carrier so torch can alias plan-owned device memory."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


def _view(ptr, shape, dtype, device):
    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.int16: "<i2"}[dtype]
    with L.device_guard(device):
        return torch.as_tensor(_DevMem(ptr, shape, typestr), device=device)


# ------------------------------------------------------------------------------------------------
# dense BEV head on the matrix cores (csrc/dense_conv.hip)
# ------------------------------------------------------------------------------------------------
def split_planes_like(b, h, w, c, device):
    """Two 16-bit NHWC planes ("hi", "lo") stored as int16 (bf16 or f16 pieces by the arithmetic of their producer)."""
    both = torch.empty((2, b, h, w, c), dtype=torch.int16, device=device)  # contiguous: one fill clears both planes
    return both[0], both[1]


def _prec_struct(pr):
    """pr = None (bf16x3) | (in_entry, out_entry | None, range_flag | None): device (4,) float32 entries of an f16s call."""
    if pr is None:
        return None, None
    in_entry, out_entry, flag = pr[:3]
    w_inv = pr[3] if len(pr) > 3 else None   # (1,) device float: hot copy of the image's 1 / s_w (DenseHeadPlan keeps them in one line)
    w_inv2 = pr[4] if len(pr) > 4 else None
    st = L.Conv2dPrec(L.PREC_F16S, L.ptr(in_entry), L.ptr(out_entry), L.ptr(flag), L.ptr(w_inv), L.ptr(w_inv2))
    return C.byref(st), st  # (the structure must outlive the call: the caller keeps the second value)


def conv2d_split(x_hi, x_lo, image, bias, relu, cin, cout, ksize, out_split=True, out_nchw=False, occ=None, reach=0, bg=None, work=None,
                 out=None, tile_state=None, reset=None, pr=None):
    """One split-precision convolution on split NHWC planes; returns (y_hi, y_lo) and/or fp32 (B,cout,H,W).
    occ / reach / bg = (bg_hi, bg_lo) [/ work: 2 zeroed int32 of the caller's, see the header]: background skipping
    (v3d_conv2d_nhwc_split with occ), same values.  out = (y_hi, y_lo): write into these planes; with tile_state (one int32 per
    tile of a PERSISTENT `out`, see the header) background tiles that already hold the empty-map response are not written.
    pr: None = bf16x3 (image from pack_conv_weight(.., "bf16x3")); (in_entry, out_entry, range_flag) = f16s."""
    b, h, w, c = x_hi.shape
    assert c == cin
    dev = x_hi.device
    y_hi = y_lo = y = None
    if out_split:
        y_hi, y_lo = out if out is not None else split_planes_like(b, h, w, cout, dev)
        if tuple(y_hi.shape) != (b, h, w, cout) or tuple(y_lo.shape) != (b, h, w, cout):
            raise RuntimeError("conv2d_split: `out` planes do not match the output geometry")
    if out_nchw:
        y = torch.empty((b, cout, h, w), dtype=torch.float32, device=dev)
    skipping = occ is not None and bg is not None
    ref, keep = _prec_struct(pr)
    with L.device_guard(dev):
        # reset = (int32 tensor, words): OTHER call sites' counters this launch zeroes before its tiles run (words may be 0) instead
        # of resetting its own pair at its end (v3d_conv2d_nhwc_split; DenseHeadPlan.forward chains the layers)
        L.check(L.lib().v3d_conv2d_nhwc_split(L.ptr(x_hi), L.ptr(x_lo), L.ptr(image), L.ptr(bias), int(bool(relu)), b, h, w, cin, cout,
                                              ksize, L.ptr(y_hi), L.ptr(y_lo), L.ptr(y), L.ptr(occ) if skipping else None,
                                              int(reach) if skipping else 0, L.ptr(bg[0]) if skipping else None,
                                              L.ptr(bg[1]) if skipping else None, L.ptr(work) if skipping else None,
                                              L.ptr(tile_state) if (skipping and work is not None and out is not None) else None,
                                              (reset[0].data_ptr() if (skipping and reset is not None) else None),
                                              (int(reset[1]) if (skipping and reset is not None) else 0), ref, L.stream_ptr()),
                "conv2d_nhwc_split")
    del keep
    return (y_hi, y_lo), y


def pack_conv_weight(weight, scale=None, precision="bf16x3"):
    """(Cout,Cin,k,k) fp32 [* per-cout scale] -> packed split image (uint8 tensor) for the given arithmetic."""
    w = weight.detach().to(torch.float32).contiguous()
    cout, cin, k, _ = w.shape
    lib = L.lib()
    img = torch.empty(int(lib.v3d_conv2d_weight_image_bytes(cin, cout, k)), dtype=torch.uint8, device=w.device)
    sc = None if scale is None else scale.detach().to(torch.float32).contiguous()
    with L.device_guard(w.device):
        L.check(lib.v3d_conv2d_pack_weights(L.ptr(w), L.ptr(sc), cout, cin, k, L.PRECISIONS[precision], L.ptr(img), L.stream_ptr()),
                "conv2d_pack_weights")
    return img


def act_entry_from_tensor(x, headroom_bits=0):
    """(4,) device entry {s, 1/s, limit, max} from the EXACT maximum of a float32 tensor (v3d_act_scale_from_rows)."""
    x = L.as_f32("act_entry_from_tensor", x)
    entry = torch.empty(4, dtype=torch.float32, device=x.device)
    with L.device_guard(x.device):
        L.check(L.lib().v3d_act_scale_from_rows(L.ptr(x), None, x.numel(), 1, int(headroom_bits), L.ptr(entry),
                                                 L.ptr(L.scale_scratch(x.device)), L.stream_ptr()), "act_scale_from_rows")
    return entry


def rows_split(rows, precision="fp32", entry=None):
    """fp32 rows (n, C) -> the same rows split into the arithmetic's 16-bit pieces, (n, 2 C) int16 = [hi | lo] per row
    (v3d_sparse_rows_split): what a packed sparse layer gathers through `in_split` (f16s: pieces of x * entry[0])."""
    x = L.as_f32("rows_split", rows)
    n, c = x.shape
    out = torch.empty((n, 2 * c), dtype=torch.int16, device=x.device)
    with L.device_guard(x.device):
        L.check(L.lib().v3d_sparse_rows_split(L.ptr(x), None, max(n, 1), c, L.PRECISIONS[precision], L.ptr(entry), L.ptr(out), L.stream_ptr()),
                "sparse_rows_split")
    return out


def tag_planes(hi, entry, flag=None):
    """f16s planes know their scale: the (4,) device entry they were written under (and the frame's range-flag word) ride on the
    `hi` tensor object as attributes -- DenseHeadPlan.forward picks them up when the caller does not pass them."""
    hi.v3d_entry, hi.v3d_flag = entry, flag
    return hi


def to_split_nhwc(x, precision="bf16x3"):
    """fp32 (B,C,H,W) -> split planes (entry point for tensors that come from torch).  f16s: the planes' scale entry is taken
    from the tensor's own maximum and attached to `hi` (tag_planes)."""
    x = L.as_f32("to_split_nhwc", x)
    b, c, h, w = x.shape
    hi, lo = split_planes_like(b, h, w, c, x.device)
    f16s = L.PRECISIONS[precision] == L.PREC_F16S
    entry = act_entry_from_tensor(x) if f16s else None
    with L.device_guard(x.device):
        L.check(L.lib().v3d_nchw_to_split_nhwc(L.ptr(x), b, c, h, w, L.ptr(hi), L.ptr(lo), L.PRECISIONS[precision], L.ptr(entry),
                                                L.stream_ptr()), "nchw_to_split_nhwc")
    tag_planes(hi, entry)
    return hi, lo


def planes_abs_max(hi, lo, entry):
    """0-dim device tensor: largest magnitude held by f16s planes written under `entry` (calibration)."""
    v = hi.view(torch.float16).float() + lo.view(torch.float16).float()
    return v.abs().max() * entry[1]


class DenseHeadState(object):
    """Per stream / captured graph state of a DenseHeadPlan: the tile counters of the persistent skipping kernels, one pair of
    output planes per RPN layer that stays put from frame to frame, and per layer one word per 80-pixel tile saying whether the
    tile currently holds computed values (1) or the layer's empty-map response (0).  That response only depends on the weights
    (and, f16s, on the layer's scale entry): a background tile that was background in the frame before needs no write
    (csrc/dense_conv.hip, DcParams::tile_state)."""

    def __init__(self, plan, device):
        self.device = device
        self.counters = torch.zeros(2 * len(plan.layers), dtype=torch.int32, device=device)
        self.key, self.out, self.tiles = None, None, None

    def ensure(self, plan, b, h, w):
        geom = (int(b), int(h), int(w), tuple(ly["cout"] for ly in plan.layers[:-1]))
        key = geom + (plan._stamp, plan.calibration_generation)
        if key == self.key:
            return
        if self.out is None or self.key[:4] != geom:
            self.out = [split_planes_like(b, h, w, ly["cout"], self.device) for ly in plan.layers[:-1]]
            n = int(L.lib().v3d_conv2d_bg_tiles(int(b), int(h), int(w)))
            self.tiles = [torch.ones(n, dtype=torch.int32, device=self.device) for _ in plan.layers[:-1]]
        else:  # new weights or scales: the planes stay where they are (captured graphs hold their addresses), nothing is in place
            for t in self.tiles:
                t.fill_(1)
        self.key = key


class DenseHeadPlan(object):
    """RPN (conv+BN+ReLU stack, detector/second.py:58-94) + the two 1x1 heads (proposal.py:19-22) as 8 split-precision MFMA
    convolutions with folded BatchNorm; weights re-packed automatically when tensors change.
    precision "fp32" (f16s): every RPN layer's output planes have a static scale entry in `self.tab`, set by `calibrate` from one
    frame with headroom (automatic on the first eager forward); "bf16x3": scale-free."""

    fuse_tail = True  # False: the 1x1 up-conv and the head as two launches (cross-check of the fused kernel, tests)

    def __init__(self, rpn, head, precision="bf16x3"):
        if precision not in L.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(L.PRECISIONS)}")
        self.rpn, self.head = rpn, head
        self.precision = precision
        self._stamp = None
        self.layers = []
        self._background = {}
        self.tab = None                   # (RPN layers, 4) float32 device scale entries of the layers' output planes (f16s)
        self.calibrated = False
        self.calibration_generation = 0

    @property
    def f16s(self):
        return L.PRECISIONS[self.precision] == L.PREC_F16S

    def set_precision(self, precision):
        if precision not in L.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(L.PRECISIONS)}")
        if L.PRECISIONS[precision] != L.PRECISIONS[self.precision]:
            self.precision, self._stamp, self.calibrated = precision, None, False

    def _pairs(self):
        mods = list(self.rpn.down_block) + list(self.rpn.up_block)
        convs = [m for m in mods if isinstance(m, nn.Conv2d)]
        bns = [m for m in mods if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        assert len(convs) == len(bns)
        return list(zip(convs, bns))

    def _param_stamp(self):
        tensors = [t for c, b in self._pairs() for t in (c.weight, b.running_mean, b.running_var, b.weight, b.bias)]
        if self.head is not None:
            tensors += [self.head.conv_cls.weight, self.head.conv_cls.bias, self.head.conv_reg.weight, self.head.conv_reg.bias]
        return tuple((t.data_ptr(), t._version) for t in tensors) + (self.precision,)

    def weights_changed(self):
        """True when a module tensor changed since the last upload: the packed images (new tensors) are then stale -- a captured
        graph that baked their addresses in must be captured again (detector/graph.py)."""
        return self._stamp is not None and self._param_stamp() != self._stamp

    def sync_weights(self):
        pairs = self._pairs()
        stamp = self._param_stamp()
        if stamp == self._stamp:
            return
        layers = []
        with torch.no_grad():
            for conv, bn in pairs:
                if bn.training:
                    raise RuntimeError("DenseHeadPlan runs eval-mode BatchNorm only")
                k = conv.kernel_size[0]
                if k not in (1, 3) or conv.stride != (1, 1) or conv.bias is not None:
                    raise NotImplementedError("DenseHeadPlan: RPN convs must be 1x1/3x3, stride 1, bias-free")
                scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                shift = (bn.bias - bn.running_mean * scale).float().contiguous()
                # (calibration bound |out| <= l1 * max|in| + bmax: largest L1 norm of a folded filter, largest |bias|)
                l1 = (conv.weight.float() * scale.view(-1, 1, 1, 1)).abs().sum((1, 2, 3)).max()
                layers.append(dict(img=pack_conv_weight(conv.weight, scale, self.precision), bias=shift, relu=True,
                                   cin=conv.in_channels, cout=conv.out_channels, k=k, l1=l1, bmax=shift.abs().max()))
            if self.head is not None:
                w = torch.cat((self.head.conv_cls.weight, self.head.conv_reg.weight), 0)
                bias = torch.cat((self.head.conv_cls.bias, self.head.conv_reg.bias), 0).float().contiguous()
                layers.append(dict(img=pack_conv_weight(w, None, self.precision), bias=bias, relu=False, cin=w.shape[1], cout=w.shape[0], k=1))
            else:
                layers.append(None)  # RPN only (RPN.native_forward): no head layer
        self.layers, self._stamp = layers, stamp
        self.calibrated = False  # the scale entries belong to the old weights' activations
        self._invalidate_background()
        if self.f16s:
            dev = layers[0]["bias"].device
            if self.tab is None or self.tab.shape[0] != len(layers) - 1 or self.tab.device != dev:
                self.tab = torch.tensor([[1.0, 1.0, 32768.0, 0.0]] * (len(layers) - 1), dtype=torch.float32, device=dev)
                self.w_inv = torch.ones(len(layers), dtype=torch.float32, device=dev)
            # the images' 1 / s_w (float 1 of each 256-byte trailer, written by the pack kernels) gathered into ONE hot line: the
            # kernels would otherwise open every launch on a cold miss of a line nothing else reads
            for i, ly in enumerate(layers):
                if ly is not None:
                    self.w_inv[i:i + 1].copy_(ly["img"][-256:].view(torch.float32)[1:2])

    def _invalidate_background(self):
        """The empty-map responses belong to the old weights / scales: recomputed at the next use -- INTO the same tensors when
        they exist (captured graphs hold their addresses)."""
        for planes in self._background.values():
            planes["valid"] = False
        self.calibration_generation += 1  # (DenseHeadState: nothing is in place any more)

    def recalibrate(self):
        """f16s: the next eager forward re-derives the layers' scale entries from its own frame (after a RangeOverflow)."""
        self.calibrated = False

    def new_work(self, device):
        """Zeroed tile-counter scratch for `forward(..., work=...)`: keep one per stream / captured graph."""
        self.sync_weights()
        return torch.zeros(2 * len(self.layers), dtype=torch.int32, device=device)

    def new_state(self, device):
        """`new_work` plus PERSISTENT output planes for every RPN layer and their tile states (DenseHeadState): background tiles
        that already hold the layer's empty-map response from an earlier frame are then not written at all."""
        self.sync_weights()
        return DenseHeadState(self, device)

    def _pr(self, i, in_entry, flag, planes_out=True):
        """f16s call description of layer i: input entry = the caller's for layer 0, the previous layer's else."""
        if not self.f16s:
            return None
        src = in_entry if i == 0 else self.tab[i - 1]
        nxt = self.w_inv[i + 1:i + 2] if i + 1 < self.w_inv.numel() else None  # (the fused tail's second image: the head)
        return (src, self.tab[i] if planes_out else None, flag, self.w_inv[i:i + 1], nxt)

    def background(self, h, w, device):
        """Per RPN layer: its output on an EMPTY (all-zero) BEV map of one image, as split planes -- what every pixel far
        enough from all occupied pixels evaluates to, borders included (`forward(..., occ=...)`).  Computed once per weight
        set (f16s: and calibration), map size by the same kernels."""
        key = (int(h), int(w), str(device))
        rec = self._background.get(key)
        if rec is None or not rec["valid"]:
            x_hi, x_lo = split_planes_like(1, h, w, self.layers[0]["cin"], device)
            x_hi.zero_()
            x_lo.zero_()
            zero_entry = None
            if self.f16s:  # (an all-zero map: any scale describes it)
                zero_entry = torch.tensor([1.0, 1.0, 32768.0, 0.0], dtype=torch.float32, device=device)
            planes = []
            for i, ly in enumerate(self.layers[:-1]):
                out = None if rec is None else rec["planes"][i]
                (x_hi, x_lo), _ = conv2d_split(x_hi, x_lo, ly["img"], ly["bias"], ly["relu"], ly["cin"], ly["cout"], ly["k"], out=out,
                                               pr=self._pr(i, zero_entry, None))
                planes.append((x_hi, x_lo))
            self._background[key] = dict(planes=planes, valid=True)
        return self._background[key]["planes"]

    def calibrate(self, x_hi, x_lo, in_entry, headroom_bits=CALIB_HEADROOM_BITS):
        """f16s: set every layer's scale entry from THIS frame (device-side, no host synchronisation; never during capture).
        The chain runs once with provisional entries from the rigorous bound |out| <= L1(filter) * max|in| + max|bias| (it cannot
        overflow; it only wastes a few bits of the small values), reads each layer's true maximum off the planes it wrote and
        stores the entry for that maximum with `headroom_bits` of margin."""
        self.sync_weights()
        if not self.f16s:
            return
        amax = planes_abs_max(x_hi, x_lo, in_entry)  # of THIS frame's planes (in_entry[3] is the calibration frame's)
        src = in_entry
        for i, ly in enumerate(self.layers[:-1]):
            prov = scale_entry_from_max(ly["l1"] * amax + ly["bmax"], 0)
            (x_hi, x_lo), _ = conv2d_split(x_hi, x_lo, ly["img"], ly["bias"], ly["relu"], ly["cin"], ly["cout"], ly["k"],
                                           pr=(src, prov, None))
            amax = planes_abs_max(x_hi, x_lo, prov)
            scale_entry_from_max(amax, headroom_bits, out=self.tab[i])
            src = prov
        self.calibrated = True
        self._invalidate_background()

    def forward(self, x_hi, x_lo, want_features=False, occ=None, work=None, in_entry=None, range_flag=None):
        """split BEV planes -> fp32 head maps (B, n_cls*n_yaw*(1+DOF), H, W) [+ fp32 RPN features].
        occ: BackbonePlan.bev_occupancy() of the same frame -- tiles of the RPN layers whose receptive field holds no occupied
        BEV pixel are copied from the empty-map response instead of convolved (identical values; a sparse scene leaves more
        than half of the map in that state).
        work: (2 * RPN layers,) zeroed int32 scratch owned by the caller, one per stream in flight (`new_work()`): the tile
        counters of the persistent skipping kernels, which leave them zeroed for the next frame (a DenseHeadState's chain: the
        layers zero each other's pairs; a layer off the persistent kernels zeroes them with a fill).  None: allocated per call.
        f16s: in_entry = the (4,) device scale entry of the input planes (BackbonePlan.bev_entry(), to_split_nhwc),
        range_flag = (1,) int32 device word raised to 2 when a layer's output leaves its calibrated range (the plan's
        overflow_any()).  An uncalibrated plan calibrates itself on this frame first (eager calls only)."""
        self.sync_weights()
        if self.f16s:
            if in_entry is None:
                in_entry = getattr(x_hi, "v3d_entry", None)
                range_flag = getattr(x_hi, "v3d_flag", None) if range_flag is None else range_flag
            if in_entry is None:
                raise RuntimeError("DenseHeadPlan (f16s): the input planes carry no scale entry -- they must come from an f16s "
                                   "BackbonePlan / to_split_nhwc(x, 'fp32') (runtime.tag_planes), or pass in_entry")
            if not self.calibrated and not torch.cuda.is_current_stream_capturing():
                self.calibrate(x_hi, x_lo, in_entry)
        elif getattr(x_hi, "v3d_entry", None) is not None:
            raise RuntimeError("DenseHeadPlan (bf16x3) was handed f16s planes: plan and dense head must use one arithmetic")
        if occ is not None and work is None:
            work = self.new_work(x_hi.device)
        state = work if isinstance(work, DenseHeadState) else None
        if state is not None:
            state.ensure(self, *x_hi.shape[:3])
            work = state.counters
        feats = None
        bg = self.background(x_hi.shape[1], x_hi.shape[2], x_hi.device) if occ is not None else None
        reach = 0
        head = self.layers[-1]
        # the RPN's 1x1 up-conv and the 1x1 head on top of it as ONE pass over the pixels (v3d_conv2d_1x1_head_fused: same bits as the
        # two launches) whenever nobody asks for the RPN features themselves
        fuse_tail = (self.fuse_tail and not want_features and head is not None and len(self.layers) >= 2
                     and self.layers[-2]["k"] == 1 and self.layers[-2]["cin"] == 128 and self.layers[-2]["cout"] == 128
                     and head["k"] == 1 and head["cin"] == 128 and head["cout"] <= 16)
        # Tile counters of a persistent state: the layers reset each other's pairs (the first launched layer zeroes the pairs of all the
        # others -- left by the previous frame --, the second the first's) instead of every launch ending on the atomic round trip
        # of a self-resetting pair; needs at least two launched skipping layers
        n_skip = len(self.layers) - 1 - (1 if fuse_tail else 0)
        chain = state is not None and occ is not None and n_skip >= 2
        for i, ly in enumerate(self.layers[:-1]):
            last = i == len(self.layers) - 2
            if last and fuse_tail:
                b, h, w, _ = x_hi.shape
                maps = torch.empty((b, head["cout"], h, w), dtype=torch.float32, device=x_hi.device)
                ref, keep = _prec_struct(self._pr(i, in_entry, range_flag))
                with L.device_guard(x_hi.device):
                    L.check(L.lib().v3d_conv2d_1x1_head_fused(L.ptr(x_hi), L.ptr(x_lo), L.ptr(ly["img"]), L.ptr(ly["bias"]),
                                                               int(bool(ly["relu"])), L.ptr(head["img"]), L.ptr(head["bias"]),
                                                               int(bool(head["relu"])), b, h, w, 128, head["cout"], L.ptr(maps),
                                                               ref, L.stream_ptr()), "conv2d_1x1_head_fused")
                del keep
                return maps
            reach += ly["k"] // 2
            (x_hi, x_lo), f = conv2d_split(x_hi, x_lo, ly["img"], ly["bias"], ly["relu"], ly["cin"], ly["cout"], ly["k"],
                                           out_split=True, out_nchw=want_features and last, occ=occ, reach=reach,
                                           bg=None if bg is None else bg[i], work=None if work is None else work[2 * i:2 * i + 2],
                                           out=None if state is None else state.out[i],
                                           tile_state=None if state is None else state.tiles[i],
                                           reset=None if not chain else ((work[2:], work.numel() - 2) if i == 0 else (work, 2 if i == 1 else 0)),
                                           pr=self._pr(i, in_entry, range_flag))
            feats = f if last else feats
        ly = self.layers[-1]
        if ly is None:
            return (None, feats)
        n_rpn = len(self.layers) - 1
        _, maps = conv2d_split(x_hi, x_lo, ly["img"], ly["bias"], ly["relu"], ly["cin"], ly["cout"], ly["k"],
                               out_split=False, out_nchw=True, pr=self._pr(n_rpn, in_entry, range_flag, planes_out=False))
        return (maps, feats) if want_features else maps
