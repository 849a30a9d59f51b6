"""SECOND detector (interface of vision3d/detector/second.py:10-94): vfe -> sparse cnn -> dense RPN -> head.

`Second(cfg).forward(item)` adds P_cls (B,1,2,200,176) / P_reg (B,1,2,200,176,7);
`.inference(item)` returns (boxes, batch_idx, class_idx, scores).  `item` is the dict produced by
vision3d_amd.core.Preprocessor (same keys as the reference's).  Parameter tree = the reference's
(vfe | cnn.blocks.* | rpn.down_block.* / rpn.up_block.* | head.conv_cls / head.conv_reg).
"""
import logging
import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm

from .. import spconv
from .layers import VoxelFeatureExtractor
from .proposal import ProposalLayer
from .sparse_cnn import SpMiddleFHD


class Middle(SpMiddleFHD):
    """Sparse backbone straight to the BEV map (skips the metric-coordinate outputs)."""

    native_train = True  # training: the whole backbone as one native call each way (runtime.PlanTrainFunction)

    def _train_plan(self, n_voxels, batch_size, device):
        from ..runtime import BackbonePlan, PlanCache
        cap = 1 << max(14, (max(n_voxels, 1) - 1).bit_length())  # voxel capacity min(points, B * MAX_VOXELS) >= n_voxels
        key = (str(device), int(batch_size), cap)
        plans = self.__dict__.setdefault("_train_plans", PlanCache())
        if key not in plans:
            plans[key] = BackbonePlan(self, self.cfg, max_batch=int(batch_size), max_points=max(cap, int(batch_size) * 16384),
                                      device=device, growth=self.__dict__.get("plan_growth", 2.0))
        return plans[key]

    def forward(self, features, coordinates, batch_size):
        if self.training and torch.is_grad_enabled():
            if (self.native_train and features.is_cuda and features.dtype == torch.float32 and features.shape[0] > 0
                    and coordinates.dtype == torch.int32):
                from ..runtime import PlanTrainFunction
                plan = self._train_plan(features.shape[0], batch_size, features.device)
                if plan.train_supported():
                    # under bf16 autocast the RPN's first convolution would cast (and, in channels_last, re-lay-out) the
                    # BEV map: the plan writes it in that form directly
                    bf16 = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
                    return PlanTrainFunction.apply(plan, features, coordinates, batch_size, bf16, *plan.train_parameters())
        x = spconv.SparseConvTensor(features, coordinates.int(), self.grid_shape, batch_size)
        if self.training and torch.is_grad_enabled():
            spconv.prebuild_rulebooks(self.blocks, x)  # every host read of the step happens here, before any conv is enqueued
        return self.to_bev(self.blocks(x))


class RPN(nn.Module):
    """One-stage dense RPN: ZeroPad+Conv3x3, `blocks` x Conv3x3 (pad 1), then a `stride`x`stride` conv
    (1x1 at stride 1), every conv bias-free and followed by BatchNorm2d(eps 1e-3, mom 0.01) + ReLU."""

    precision = os.environ.get("V3D_PRECISION", "fp32")  # arithmetic of native_forward (see Second.precision)

    def __init__(self, C_in=128, C_up=128, C_down=128, blocks=5):
        super().__init__()
        self.down_block, C_in = self._make_down_block(C_in, C_down, blocks)
        self.up_block = self._make_up_block(C_in, C_up)
        self._init_weights()

    @staticmethod
    def _bn(planes):
        return nn.BatchNorm2d(planes, eps=1e-3, momentum=0.01)

    def _make_down_block(self, inplanes, planes, num_blocks, stride=1):
        layers = [nn.ZeroPad2d(1), nn.Conv2d(inplanes, planes, 3, stride=stride, bias=False), self._bn(planes), nn.ReLU()]
        for _ in range(num_blocks):
            layers += [nn.Conv2d(planes, planes, 3, padding=1, bias=False), self._bn(planes), nn.ReLU()]
        return nn.Sequential(*layers), planes

    def _make_up_block(self, inplanes, planes, stride=1):
        return nn.Sequential(nn.Conv2d(inplanes, planes, stride, stride=stride, bias=False), self._bn(planes), nn.ReLU())

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
            elif isinstance(m, _BatchNorm):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            if x.is_cuda:  # inference on the GPU: the bf16x3 MFMA convolutions of csrc/dense_conv.hip, never torch / MIOpen
                return self.native_forward(x)
            return self.fused_forward(x)  # host tensors (the CPU golden tests of the module arithmetic)
        mods = list(self.down_block)
        if (isinstance(mods[0], nn.ZeroPad2d) and isinstance(mods[1], nn.Conv2d) and mods[1].padding == (0, 0)
                and mods[1].padding_mode == "zeros" and mods[0].padding == (1, 1, 1, 1)):
            # ZeroPad2d(1) + conv(padding 0) == conv(padding 1): same arithmetic without materialising the padded map
            # (a 144 MB copy forwards and one backwards at bs = 8)
            c = mods[1]
            x = F.conv2d(x, c.weight, c.bias, c.stride, 1, c.dilation, c.groups)
            for m in mods[2:]:
                x = m(x)
            return self.up_block(x)
        return self.up_block(self.down_block(x))

    def _folded(self):
        """Eval-mode BatchNorm folded into each conv's weight/bias, cached until a tensor changes."""
        convs = [m for m in list(self.down_block) + list(self.up_block) if isinstance(m, nn.Conv2d)]
        bns = [m for m in list(self.down_block) + list(self.up_block) if isinstance(m, _BatchNorm)]
        stamp = tuple((t.data_ptr(), t._version) for c, b in zip(convs, bns)
                      for t in (c.weight, b.running_mean, b.running_var, b.weight, b.bias))
        cache = self.__dict__.get("_fold_cache")
        if cache is None or cache[0] != stamp:
            folded = []
            with torch.no_grad():
                for c, b in zip(convs, bns):
                    inv = torch.rsqrt(b.running_var + b.eps) * b.weight
                    folded.append(((c.weight * inv.view(-1, 1, 1, 1)).contiguous(), (b.bias - b.running_mean * inv).contiguous(),
                                   c.padding))
            cache = (stamp, folded)
            self.__dict__["_fold_cache"] = cache
        return cache[1]

    def native_forward(self, x):
        """fp32 (B, C, H, W) cuda -> fp32 (B, C_up, H, W): the seven conv + folded-BN + ReLU layers on csrc/dense_conv.hip
        (split bf16 NHWC planes in between).  What `model.rpn(bev)` runs in eval mode on the GPU."""
        from ..runtime import DenseHeadPlan, to_split_nhwc
        plan = self.__dict__.get("_native_plan")
        if plan is None:
            plan = self.__dict__["_native_plan"] = DenseHeadPlan(self, None, precision=self.precision)
        plan.set_precision(self.precision)
        if not plan.f16s:
            hi, lo = to_split_nhwc(x.float())
            return plan.forward(hi, lo, want_features=True)[1]
        # f16s: the input's scale entry comes from the tensor itself, the layers' entries from a calibration on EVERY call (a
        # module-level call has no frame stream to amortise one over, and no flag reader): exact maxima, nothing can leave the range
        hi, lo = to_split_nhwc(x.float(), plan.precision)
        plan.calibrate(hi, lo, hi.v3d_entry, headroom_bits=0)
        return plan.forward(hi, lo, want_features=True)[1]

    def fused_forward(self, x):
        """Host path (CPU tensors): 7 x (conv with folded BN, in-place ReLU)."""
        folded = self._folded()
        x = F.pad(x, (1, 1, 1, 1))
        for w, b, pad in folded:
            x = F.conv2d(x, w, b, padding=pad).relu_()
        return x


_log = logging.getLogger("vision3d_amd")


class Second(nn.Module):
    # native inference paths: RPN tiles far from every occupied BEV pixel are copied from the empty-map response instead of
    # convolved (runtime.DenseHeadPlan.forward(occ=...)); the values are identical, False only for A/B measurements
    skip_background = os.environ.get("V3D_SKIP_BACKGROUND", "1") != "0"
    # arithmetic of the native INFERENCE paths (sparse backbone plan + dense head): "fp32" = f16 hi / lo pieces under calibrated
    # power-of-two scales -- the reference's fp32 modules up to summation noise, same MFMA count as "bf16x3" (bf16 pieces, 2^-17 per
    # product, no calibration): csrc/spconv.hip "the split-precision product".  Set before the first inference call.
    precision = os.environ.get("V3D_PRECISION", "fp32")

    def __init__(self, cfg):
        super().__init__()
        self.vfe = VoxelFeatureExtractor()
        self.cnn = Middle(cfg)
        # BEV channels = 64 x (z extent after the last sparse stage): 128 for the KITTI grid, as hard-coded in
        # the reference (second.py:52); other ranges (e.g. the Waymo-range stress config: 3 z-slices) follow.
        z = self.cnn.grid_shape[0]
        for k, st, pd in ((3, 2, 1), (3, 2, 1), (3, 2, 0), (3, 2, 0)):
            z = (z + 2 * pd - k) // st + 1
        self.rpn = RPN(C_in=64 * z)
        self.head = ProposalLayer(cfg)
        self.cfg = cfg
        # "the parameters may have changed" counter for the captured-graph runners (detector/graph.py): bumped by load_state_dict, by
        # every train() / eval() switch (an optimizer step happens in training mode) and by notify_weights_changed() (in-place edits
        # in eval mode).  A runner compares this ONE integer per launch -- comparing the ~120 parameter version counters per frame
        # cost 0.25 ms of host time per frame (profiles/r06_b_*: 3 800 -> 2 900 frames/s) -- and looks at the tensors only when it moved.
        self.__dict__["_weights_epoch"] = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: (module.notify_weights_changed(), None)[1])

    def notify_weights_changed(self):
        """Tell the captured-graph runners that parameters / buffers were edited (they check the tensors themselves on their next
        launch).  Called automatically by load_state_dict and train() / eval(); call it after in-place edits of an eval-mode model.
        The eager entry points need no notice: they compare the tensors' version counters on every call."""
        self.__dict__["_weights_epoch"] = self.__dict__.get("_weights_epoch", 0) + 1
        return self

    def train(self, mode=True):
        self.notify_weights_changed()
        return super().train(mode)

    def set_precision(self, precision):
        """Arithmetic of every native INFERENCE path of this model: the plans (Second.precision), RPN.native_forward and the
        op-by-op sparse modules.  "fp32" (default) or "bf16x3"."""
        from .. import _lib as L
        if precision not in L.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(L.PRECISIONS)}")
        self.precision = self.rpn.precision = precision
        for m in self.modules():
            if isinstance(m, spconv.conv._SparseConvBase):
                m.precision = precision
        return self

    def feature_extract(self, item):
        if "voxel_mean" in item:  # fused into the device voxelizer
            features = item["voxel_mean"]
        else:
            features = self.vfe(item["features"], item["occupancy"])
        features = self.cnn(features, item["coordinates"], item["batch_size"])
        return self.rpn(features)

    # ---- the reference's entry points (train.py:63 calls forward(item), inference.py:38 calls inference(item)).  In eval mode
    #      without autograd an item of the device Preprocessor runs the native path: backbone plan fed with the item's voxels
    #      (csrc/second_plan.hip) -> bf16x3 MFMA RPN + heads (csrc/dense_conv.hip) -> device proposal stage (csrc/proposal.hip).
    #      Training (autograd) keeps the module-by-module path.
    def _native_item(self, item):
        if self.training or torch.is_grad_enabled() or "voxel_mean" not in item:
            return False
        vm, co = item["voxel_mean"], item["coordinates"]
        return vm.is_cuda and co.is_cuda and co.dtype == torch.int32 and vm.dtype == torch.float32 and vm.shape[0] > 0

    def _head_maps_from_item(self, item):
        """-> (fused [cls | reg] head maps (B, n_anchor * (1 + DOF), H, W) fp32, the plan that produced them)."""
        m, b = item["voxel_mean"].shape[0], int(item["batch_size"])
        cap_pts = 1 << max(14, (max(m, 1) - 1).bit_length())  # the plan's voxel capacity is min(points, B * MAX_VOXELS) >= M
        plan = self.backbone_plan(b, max(cap_pts, b * 16384))
        hi, lo = plan.forward_voxels_split(item["voxel_mean"], item["coordinates"], b)
        return self.dense_plan().forward(hi, lo, occ=plan.bev_occupancy(b) if self.skip_background else None,
                                         in_entry=plan.bev_entry(), range_flag=plan.overflow_any()), plan

    def _with_recalibration(self, run, plan_of):
        """`run()` once; when the frame left the range the f16s scale entries were calibrated for (runtime.RangeOverflow: the
        frame's summary word, read in the frame's one host synchronisation), recalibrate on THIS frame and run it again."""
        from ..runtime import RangeOverflow
        try:
            return run()
        except RangeOverflow:
            plan_of().recalibrate()
            self.dense_plan().recalibrate()
            out = run()
            self.spread_calibration(plan_of())
            return out

    # training: RPN + heads forward / backward on csrc/dense_train.hip instead of torch / MIOpen, in the arithmetic of the caller's step:
    #   * under bf16 autocast (what bench.py --mode train does, BASELINE configs[2]): bf16 storage, fp32 accumulation -- the contract
    #     of `torch.autocast("cuda", torch.bfloat16)`;
    #   * outside autocast -- the reference's fp32 script, train.py:58-66 -- `dense_train_precision` (V3D_DENSE_TRAIN_PRECISION):
    #       "bf16x3" (default)  the fp32-class step: split hi + lo storage, three-term products (dense_train.py): no MIOpen
    #                           convolution anywhere in the step, nothing to opt into;
    #       "bf16"              the model enters bf16 autocast ITSELF for the dense half (the faster, reduced-precision step);
    #       "torch"             the torch modules (MIOpen fp32 convolutions), as the reference runs them;
    #       "fp32"              (the value rounds 1-4 documented for the torch modules) = "torch".
    native_dense_train = os.environ.get("V3D_DENSE_TRAIN", "native") == "native"
    dense_train_precision = os.environ.get("V3D_DENSE_TRAIN_PRECISION", "bf16x3")

    def _train_head_maps(self, item):
        """-> fused fp32 head maps of a TRAINING forward through the native dense plan [or the (scores, boxes) pair of the torch
        modules where the plan does not apply], or None when the torch modules are asked for (see above)."""
        if not (self.native_dense_train and self.training and torch.is_grad_enabled()):
            return None
        autocast = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
        if autocast:
            return self._train_head_maps_autocast(item)
        if torch.is_autocast_enabled("cuda"):  # (another autocast dtype: the torch modules know what to do with it)
            return None
        if self.dense_train_precision == "fp32":  # the pre-round-5 name of the torch-module step: still honoured
            return None
        if self.dense_train_precision == "bf16":
            with torch.autocast("cuda", dtype=torch.bfloat16):  # opted in: the model enters the contract itself
                return self._train_head_maps_autocast(item)
        if self.dense_train_precision == "bf16x3":
            return self._train_head_maps_split(item)
        if self.dense_train_precision != "torch":
            raise ValueError(f"dense_train_precision must be 'bf16x3', 'bf16', 'torch' or 'fp32' (= 'torch'), not {self.dense_train_precision!r}")
        return None

    def _train_head_maps_split(self, item):
        from .. import dense_train
        features = item["voxel_mean"] if "voxel_mean" in item else self.vfe(item["features"], item["occupancy"])
        bev = self.cnn(features, item["coordinates"], item["batch_size"])
        if not dense_train.supported(self.rpn, self.head, bev, "bf16x3"):
            return self._torch_dense_fallback(bev, dense_train.why_unsupported(self.rpn, self.head, bev, "bf16x3"))
        return dense_train.train_head_maps(self.rpn, self.head, bev, self.__dict__.setdefault("_dense_train_plans", {}), "bf16x3")

    def _train_head_maps_autocast(self, item):
        from .. import dense_train
        features = item["voxel_mean"] if "voxel_mean" in item else self.vfe(item["features"], item["occupancy"])
        bev = self.cnn(features, item["coordinates"], item["batch_size"])
        if not bev.is_cuda or bev.dim() != 4 or bev.shape[1] != 128:
            return self._torch_dense_fallback(bev, "BEV map is not a 128-channel CUDA tensor")
        if bev.dtype != torch.bfloat16 or not bev.is_contiguous(memory_format=torch.channels_last):
            bev = bev.to(dtype=torch.bfloat16, memory_format=torch.channels_last)  # what autocast's first conv would do
        if not dense_train.supported(self.rpn, self.head, bev):
            return self._torch_dense_fallback(bev, dense_train.why_unsupported(self.rpn, self.head, bev))
        return dense_train.train_head_maps(self.rpn, self.head, bev, self.__dict__.setdefault("_dense_train_plans", {}))

    # the training forward leaves the native dense kernels for the torch modules (MIOpen): counted, and logged ONCE per model
    torch_dense_fallbacks = 0

    def _torch_dense_fallback(self, bev, why):
        self.torch_dense_fallbacks += 1
        if self.torch_dense_fallbacks == 1:
            _log.warning("Second.forward (training): dense RPN / head run through the torch modules, not csrc/dense_train.hip: %s", why)
        return self.head(self.rpn(bev))

    def check_train_overflow(self):
        """Capacity check of the last training forward without a stall (runtime.BackbonePlan.check_deferred_overflow): call it
        between backward() and optimizer.step(), and once after the last step of a run."""
        for plan in self.cnn.__dict__.get("_train_plans", {}).values():
            plan.check_deferred_overflow()

    def forward(self, item):
        # the fused maps of an EARLIER training forward must never reach ProposalLoss through a re-used item dict
        item.pop("_head_maps", None)
        if self._native_item(item):
            state = {}

            def run():
                maps, state["plan"] = self._head_maps_from_item(item)
                state["plan"].check_overflow()  # no count is read on this path: one blocking word
                return maps
            scores, boxes = self.head.maps_from_fused(self._with_recalibration(run, lambda: state["plan"]))
        elif (maps := self._train_head_maps(item)) is not None:
            scores, boxes = maps if isinstance(maps, tuple) else self.head.maps_from_fused(maps)
            if not isinstance(maps, tuple):
                # ProposalLoss takes its native pass on the fused maps (detector/proposal.py) -- only while P_cls / P_reg are
                # still the views made here (it compares identities): post-processed outputs go through the torch expressions
                item["_head_maps"] = (maps, scores, boxes)
        else:
            scores, boxes = self.head(self.feature_extract(item))
        item.update(dict(P_cls=scores, P_reg=boxes))
        return item

    def inference(self, item):
        if self._native_item(item):
            state = {}

            def run():
                maps, state["plan"] = self._head_maps_from_item(item)
                return self.head.inference_native(maps, item["anchors"], overflow_flag=state["plan"].overflow_any())
            return self._with_recalibration(run, lambda: state["plan"])
        return self.head.inference(self.feature_extract(item), item["anchors"])

    # ---- fused path: raw device points in, proposals out (voxelizer + sparse backbone in one native call)
    def backbone_plan(self, max_batch, max_points, slot=0):
        """Plans own their arena (hash tables, rulebooks, per-layer outputs): frames in flight at the same time need
        one plan each -- `slot` keys them."""
        from ..runtime import BackbonePlan, PlanCache
        plans = self.__dict__.setdefault("_plans", PlanCache())
        dev = next(self.parameters()).device
        key = (str(dev), int(max_batch), int(max_points)) + ((int(slot),) if slot else ())
        if key not in plans:
            plans[key] = BackbonePlan(self.cnn, self.cfg, max_batch=max_batch, max_points=max_points, device=dev,
                                      growth=self.__dict__.get("plan_growth", 2.0), precision=self.precision)
        plans[key].set_precision(self.precision)
        self.share_calibration(plans[key])
        return plans[key]

    def share_calibration(self, plan):
        """f16s: the plans of one model (the slots of a pipeline, different batch capacities) use ONE set of scale entries -- a plan
        that has none yet takes them from a calibrated peer instead of calibrating on whatever frame it sees first, so every slot
        computes the same bits for the same frame."""
        if not plan.f16s or plan._calib == "done":
            return
        for other in self.__dict__.get("_plans", {}).values():
            if other is not plan and other.f16s and other._calib == "done" and other.device == plan.device:
                plan.copy_calibration(other)
                return

    def spread_calibration(self, plan):
        """... and after `plan` was recalibrated (a frame left the range), its peers follow."""
        for other in self.__dict__.get("_plans", {}).values():
            if other is not plan and other.f16s and plan.f16s and other.device == plan.device:
                other.copy_calibration(plan)

    def bev_from_points(self, clouds):
        """clouds: list of (N_b, C) float32 cuda tensors -> BEV map (B, 128, 200, 176); eval mode only."""
        offsets = [0]
        for c in clouds:
            offsets.append(offsets[-1] + int(c.shape[0]))
        flat = clouds[0] if len(clouds) == 1 else torch.cat(clouds, dim=0)
        cap_pts = 1 << max(14, (offsets[-1] - 1).bit_length())  # round the point capacity up: few distinct plans
        plan = self.backbone_plan(len(clouds), cap_pts)
        return plan.forward(flat, offsets)

    def _plan_for(self, clouds):
        offsets = [0]
        for c in clouds:
            offsets.append(offsets[-1] + int(c.shape[0]))
        flat = clouds[0] if len(clouds) == 1 else torch.cat(clouds, dim=0)
        cap_pts = 1 << max(14, (offsets[-1] - 1).bit_length())
        return self.backbone_plan(len(clouds), cap_pts), flat, offsets

    def dense_plan(self):
        from ..runtime import DenseHeadPlan
        if "_dense_plan" not in self.__dict__:
            self.__dict__["_dense_plan"] = DenseHeadPlan(self.rpn, self.head, precision=self.precision)
        self.__dict__["_dense_plan"].set_precision(self.precision)
        return self.__dict__["_dense_plan"]

    def head_maps_from_points(self, clouds):
        """Fully native feature path: raw points -> (cls_map, reg_map); the BEV map never leaves the split
        bf16 NHWC format between the sparse backbone and the 8 MFMA convolutions."""
        return self.head.maps_from_fused(self.fused_head_from_points(clouds))

    def fused_head_from_points(self, clouds):
        """raw points -> (B, n_anchor*(1+DOF), H, W) fp32: the [cls | reg] output of the fused 1x1 head."""
        plan, flat, offsets = self._plan_for(clouds)

        def run():
            hi, lo = plan.forward_split(flat, offsets)
            maps = self.dense_plan().forward(hi, lo, occ=plan.bev_occupancy(len(clouds)) if self.skip_background else None,
                                             in_entry=plan.bev_entry(), range_flag=plan.overflow_any())
            if plan.f16s:  # (this path reads nothing else back: one blocking word, as Second.forward does)
                plan.check_overflow()
            return maps
        return self._with_recalibration(run, lambda: plan)

    def graphed_inference(self, anchors, frame_sizes):
        """Capture raw points -> candidates+NMS as ONE HIP graph for a fixed batch geometry (frame_sizes = points
        per frame).  Returns `run(clouds) -> (boxes, batch_idx, class_idx, scores)`; each call copies the clouds
        into a static buffer, replays the graph (a single launch for ~150 kernels) and does the final
        variable-length selection.  Re-capture (call again) when the geometry changes."""
        from .graph import GraphedSecond
        return GraphedSecond(self, anchors, frame_sizes)

    def pipelined_inference(self, anchors, frame_sizes, depth=2, autotune=False):
        """Throughput mode: `depth` captured graphs on `depth` streams, frame i+1 is submitted before frame i's result
        is collected (see detector/graph.py:PipelinedSecond).  `run.submit(clouds)`, `run.collect()`.
        autotune: `depth` is the maximum; streams and depth are picked by measurement on the first frame."""
        from .graph import PipelinedSecond
        return PipelinedSecond(self, anchors, frame_sizes, depth, autotune)

    def inference_points(self, clouds, anchors, dense="mfma", proposals="native"):
        """Same result as `inference(Preprocessor(cfg)(...))` without materialising the intermediate dict.  RPN + heads run on
        the hand-written bf16x3 MFMA convolution (csrc/dense_conv.hip) -- there is no torch / MIOpen inference path.
        proposals = "native": top-k / decode / NMS / score cut in csrc/proposal.hip; "torch": the op-by-op
        statement of proposal.py:61-80 (ties in the top-k are then torch.topk's)."""
        if dense != "mfma":
            raise ValueError("inference_points: the torch (MIOpen) dense path was removed; dense must be 'mfma'")
        if proposals == "native":
            plan, flat, offsets = self._plan_for(clouds)

            def run():
                hi, lo = plan.forward_split(flat, offsets)
                occ = plan.bev_occupancy(len(clouds)) if self.skip_background else None
                maps = self.dense_plan().forward(hi, lo, occ=occ, in_entry=plan.bev_entry(), range_flag=plan.overflow_any())
                return self.head.inference_native(maps, anchors, overflow_flag=plan.overflow_any())
            return self._with_recalibration(run, lambda: plan)
        cls_map, reg_map = self.head_maps_from_points(clouds)
        return self.head.inference_from_maps(cls_map, reg_map, anchors)
