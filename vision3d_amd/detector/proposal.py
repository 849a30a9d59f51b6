"""BEV proposal head and its loss (interface of vision3d/detector/proposal.py:10-141).

ProposalLayer: 1x1 class/box heads -> (B, n_cls, n_yaw, ny, nx[, 7]); inference = sigmoid, top-k per
(frame, class), VoxelNet decode, multi-class rotated NMS (IoU 0.01) and per-class score threshold.
Index bookkeeping tensors are created on the scores' device (the reference mixes CPU and GPU tensors,
legal only in torch 1.4 -- SURVEY.md H9).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib as L
from ..core.box_encode import decode
from ..ops import batched_nms_rotated, batched_nms_rotated_padded, sigmoid_focal_loss


_PINNED = {}


def _pinned_pair(device):
    """Two pinned int32 words per device for the per-frame count / overflow read (host code is sequential: copy, sync, read)."""
    key = str(device)
    if key not in _PINNED:
        _PINNED[key] = torch.empty(2, dtype=torch.int32).pin_memory()
    return _PINNED[key]


def _raise_on_flag(word):
    """The frame's summary word (csrc/v3d_internal.h): 1 = a capacity was hit, 2 = an f16s tensor left its calibrated range, 3 = an
    f16s tensor stayed 2^12 below it (precision at risk: recalibrate downward)."""
    if word == 3:
        from ..runtime import RangeUnderflow
        raise RangeUnderflow("f16s arithmetic: a tensor of this frame stayed 2^12 below its calibrated range (the fp32-class precision "
                             "is not guaranteed); recalibrate on this frame and run it again (Second.inference and the graph runners do)")
    if word == 2:
        from ..runtime import RangeOverflow
        raise RangeOverflow("f16s arithmetic: a tensor of this frame exceeded its calibrated range (results invalid); recalibrate "
                            "on this frame and run it again (Second.inference and the graph runners do)")
    if word > 0:
        raise RuntimeError("sparse backbone: a stage exceeded its active-site capacity (rows were dropped); build the "
                           "plan with a larger `growth` (Second.plan_growth / BackbonePlan(growth=...))")


class ProposalLayer(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        n_anchor = cfg.NUM_CLASSES * cfg.NUM_YAW
        self.conv_cls = nn.Conv2d(cfg.PROPOSAL.C_IN, n_anchor, 1)
        self.conv_reg = nn.Conv2d(cfg.PROPOSAL.C_IN, n_anchor * cfg.BOX_DOF, 1)
        self.TOPK, self.DOF = cfg.PROPOSAL.TOPK, cfg.BOX_DOF
        self._init_weights()

    def _init_weights(self):
        # focal-prior bias exactly as the reference writes it: +1.005 (proposal.py:27)
        nn.init.constant_(self.conv_cls.bias, (-math.log(1 - .01) / .01))
        nn.init.constant_(self.conv_reg.bias, 0)
        nn.init.normal_(self.conv_cls.weight, std=0.01)
        nn.init.normal_(self.conv_reg.weight, std=0.01)

    def _generate_group_idx(self, B, n_cls, device=None):
        """(batch_idx, class_idx, group_idx), each (B * n_cls * TOPK,), group = class + n_cls * batch."""
        b = torch.arange(B, device=device).view(B, 1, 1).expand(B, n_cls, self.TOPK)
        c = torch.arange(n_cls, device=device).view(1, n_cls, 1).expand(B, n_cls, self.TOPK)
        return b.reshape(-1), c.reshape(-1), (c + n_cls * b).reshape(-1)

    def _above_score_thresh(self, scores, class_idx):
        thresh = scores.new_tensor([a["score_thresh"] for a in self.cfg.ANCHORS])
        return scores > thresh[class_idx]

    def _multiclass_batch_nms(self, boxes, scores):
        B, n_cls = scores.shape[:2]
        scores = scores.reshape(-1)
        boxes = boxes.reshape(-1, self.DOF)
        bev = boxes[:, [0, 1, 3, 4, 6]].contiguous()
        batch_idx, class_idx, group_idx = self._generate_group_idx(B, n_cls, scores.device)
        keep = batched_nms_rotated(bev, scores, group_idx, iou_threshold=0.01)
        boxes, batch_idx, class_idx, scores = (x[keep] for x in (boxes, batch_idx, class_idx, scores))
        mask = self._above_score_thresh(scores, class_idx)
        return [x[mask] for x in (boxes, batch_idx, class_idx, scores)]

    def _decode(self, reg_map, anchors, anchor_idx):
        B, n_cls = reg_map.shape[:2]
        gidx = anchor_idx[..., None].expand(-1, -1, -1, self.DOF)
        deltas = reg_map.reshape(B, n_cls, -1, self.DOF).gather(2, gidx)
        anc = anchors.reshape(1, n_cls, -1, self.DOF).expand(B, -1, -1, -1).gather(2, gidx)
        return decode(deltas, anc)

    def inference(self, feature_map, anchors):
        """-> (boxes (K,7), batch_idx (K,), class_idx (K,), scores (K,)), by decreasing score."""
        cls_map, reg_map = self(feature_map)
        return self.inference_from_maps(cls_map, reg_map, anchors)

    def maps_from_fused(self, maps):
        """(B, n_anchor*(1+DOF), H, W) = [cls | reg] channels of the fused 1x1 head -> (cls_map, reg_map)."""
        n_anchor = self.cfg.NUM_CLASSES * self.cfg.NUM_YAW
        return self.reshape_cls(maps[:, :n_anchor].contiguous()), self.reshape_reg(maps[:, n_anchor:].contiguous())

    def proposals_padded(self, cls_map, reg_map, anchors):
        """Everything up to (and including) NMS with NO host synchronisation: returns the B*n_cls*TOPK
        candidates (boxes, batch_idx, class_idx, scores) plus (keep padded, n_keep) on the device.  This is
        the part that can live inside a captured HIP graph; `finalize` does the variable-length selection."""
        score_map = cls_map.sigmoid()
        B, n_cls = score_map.shape[:2]
        scores, anchor_idx = score_map.reshape(B, n_cls, -1).topk(self.TOPK, -1)
        boxes = self._decode(reg_map, anchors, anchor_idx).reshape(-1, self.DOF)
        scores = scores.reshape(-1)
        batch_idx, class_idx, group_idx = self._generate_group_idx(B, n_cls, scores.device)
        # column select without a host-built index tensor (an H2D copy is illegal during graph capture)
        bev = torch.stack((boxes[:, 0], boxes[:, 1], boxes[:, 3], boxes[:, 4], boxes[:, 6]), dim=1)
        keep, n_keep = batched_nms_rotated_padded(bev, scores, group_idx, 0.01)
        return boxes, batch_idx, class_idx, scores, keep, n_keep

    def native_supported(self, batch_size, anchors_per_class=None):
        """csrc/proposal.hip sorts the candidates of all (frame, class) groups in one workgroup (<= 1024 of them) and
        selects each group's top-k in two levels of <= 40 register-resident slices of <= 8192 anchors."""
        ok = batch_size * self.cfg.NUM_CLASSES * self.TOPK <= 1024 and self.DOF == 7 and self.cfg.NUM_CLASSES <= 16
        if anchors_per_class is not None:
            ok = ok and anchors_per_class <= 8192 * min(40, 4096 // self.TOPK)
        return ok

    def native_proposals(self, head_maps, anchors, overflow_flag=None, host_out=None):
        """The whole stage after the 1x1 heads in libvision3d_hip.so (csrc/proposal.hip: 8 launches, no host
        sync): head_maps (B, n_anchor*(1+DOF), H, W) = [cls | reg] channels of the fused head.  Returns padded
        (boxes (N,7), batch_idx, class_idx, scores, n_out int32 on the device); capturable in a HIP graph.
        overflow_flag (a plan's (1,) device word): copied into n_out[1] by the last kernel, so that `finalize_native` reads
        the count and the plan's capacity verdict with ONE 8-byte copy.
        host_out (a PINNED int32 host tensor of two words, kept by the caller): the last kernel stores the pair straight into it
        (pinned memory is device-accessible) -- no copy is enqueued behind the frame at all; `finalize_native` synchronises the
        stream and reads it."""
        cfg = self.cfg
        L.require_gpu("proposals", head_maps, anchors)
        maps, anc = L.as_f32("proposals", head_maps), L.as_f32("proposals", anchors)
        B, ctot, H, W = maps.shape
        n_cls, n_yaw = cfg.NUM_CLASSES, cfg.NUM_YAW
        if ctot != n_cls * n_yaw * (1 + self.DOF) or self.DOF != 7 or anc.numel() != n_cls * n_yaw * H * W * 7:
            raise RuntimeError("proposals: head map / anchor grid shapes disagree")
        N = B * n_cls * self.TOPK
        dev = maps.device
        boxes = torch.empty((N, 7), dtype=torch.float32, device=dev)
        batch_idx = torch.empty((N,), dtype=torch.int64, device=dev)
        class_idx = torch.empty((N,), dtype=torch.int64, device=dev)
        scores = torch.empty((N,), dtype=torch.float32, device=dev)
        if host_out is not None:
            if not (host_out.is_pinned() and host_out.dtype == torch.int32 and host_out.numel() == 2 and overflow_flag is not None):
                raise RuntimeError("proposals: host_out is a pinned int32 pair, and needs the plan's flag word beside the count")
            n_out = host_out
        else:
            n_out = torch.empty((1 if overflow_flag is None else 2,), dtype=torch.int32, device=dev)  # written by the last kernel
        lib = L.lib()
        ws = L.workspace(lib.v3d_proposals_workspace(B, n_cls, self.TOPK), dev)
        thresh = L.host_f32([a["score_thresh"] for a in cfg.ANCHORS[:n_cls]])
        with L.device_guard(dev):
            L.check(lib.v3d_proposals_flag(L.ptr(maps), L.ptr(anc), B, n_cls, n_yaw, H, W, self.TOPK, thresh, 0.01, L.ptr(boxes),
                                           L.ptr(batch_idx), L.ptr(class_idx), L.ptr(scores), L.ptr(n_out), L.ptr(overflow_flag),
                                           L.ptr(ws), ws.numel(), L.stream_ptr()), "proposals")
        return boxes, batch_idx, class_idx, scores, n_out

    def native_topk(self, head_maps, anchors):
        """The decoded top-k candidates BEFORE NMS (v3d_proposals_topk): boxes (B, n_cls * TOPK, 7), scores (B, n_cls * TOPK), in the
        order of `scores.topk` + `_decode` of the torch statement (ties: anchor index ascending).  No host synchronisation."""
        cfg = self.cfg
        L.require_gpu("proposals_topk", head_maps, anchors)
        maps, anc = L.as_f32("proposals_topk", head_maps), L.as_f32("proposals_topk", anchors)
        B, ctot, H, W = maps.shape
        n_cls, n_yaw = cfg.NUM_CLASSES, cfg.NUM_YAW
        if ctot != n_cls * n_yaw * (1 + self.DOF) or self.DOF != 7 or anc.numel() != n_cls * n_yaw * H * W * 7:
            raise RuntimeError("proposals_topk: head map / anchor grid shapes disagree")
        N = B * n_cls * self.TOPK
        boxes = torch.empty((B, n_cls * self.TOPK, 7), dtype=torch.float32, device=maps.device)
        scores = torch.empty((B, n_cls * self.TOPK), dtype=torch.float32, device=maps.device)
        lib = L.lib()
        ws = L.workspace(lib.v3d_proposals_workspace(B, n_cls, self.TOPK), maps.device)
        with L.device_guard(maps.device):
            L.check(lib.v3d_proposals_topk(L.ptr(maps), L.ptr(anc), B, n_cls, n_yaw, H, W, self.TOPK, L.ptr(boxes), L.ptr(scores), L.ptr(ws),
                                           ws.numel(), L.stream_ptr()), "proposals_topk")
        assert N == boxes.shape[0] * boxes.shape[1]
        return boxes, scores

    def native_refine_nms(self, deltas, proposals, conf, iou_threshold=0.01, finalize=True):
        """Stage-2 tail (v3d_refine_nms): deltas / proposals (B, n_cls * TOPK, 7) and conf (B, n_cls * TOPK[, 1]) in the layout of
        `native_topk` -> (refined boxes (B, n, 7), [boxes, batch_idx, class_idx, scores] of the survivors by decreasing score).  One
        host read (the count); finalize=False: no host read -- the second value is (boxes, batch_idx, class_idx, scores, n_out) padded,
        n_out the device word (PV_RCNN.inference_end reads it through an event)."""
        cfg = self.cfg
        d, p, c = (L.as_f32("refine_nms", t) for t in (deltas, proposals, conf))
        L.require_gpu("refine_nms", d, p, c)
        B, n = p.shape[:2]
        n_cls = cfg.NUM_CLASSES
        if n != n_cls * self.TOPK or d.shape != p.shape or c.numel() != B * n or p.shape[-1] != 7:
            raise RuntimeError("refine_nms: candidates are not in the (B, n_cls * TOPK) layout of native_topk")
        N, dev = B * n, p.device
        refined = torch.empty((B, n, 7), dtype=torch.float32, device=dev)
        boxes = torch.empty((N, 7), dtype=torch.float32, device=dev)
        batch_idx = torch.empty((N,), dtype=torch.int64, device=dev)
        class_idx = torch.empty((N,), dtype=torch.int64, device=dev)
        scores = torch.empty((N,), dtype=torch.float32, device=dev)
        n_out = torch.empty((1,), dtype=torch.int32, device=dev)
        lib = L.lib()
        ws = L.workspace(lib.v3d_refine_nms_workspace(B, n_cls, self.TOPK), dev)
        thresh = L.host_f32([a["score_thresh"] for a in cfg.ANCHORS[:n_cls]])
        with L.device_guard(dev):
            L.check(lib.v3d_refine_nms(L.ptr(d), L.ptr(p), L.ptr(c), B, n_cls, self.TOPK, thresh, float(iou_threshold), L.ptr(refined),
                                       L.ptr(boxes), L.ptr(batch_idx), L.ptr(class_idx), L.ptr(scores), L.ptr(n_out), L.ptr(ws), ws.numel(),
                                       L.stream_ptr()), "refine_nms")
        if not finalize:
            return refined, (boxes, batch_idx, class_idx, scores, n_out)
        return refined, self.finalize_native(boxes, batch_idx, class_idx, scores, n_out)

    @staticmethod
    def finalize_native(boxes, batch_idx, class_idx, scores, n_out, overflow_flag=None, done=None):
        """The one host read of the frame (the reference synchronises inside its NMS): the number of proposals and, when
        the frame came through a BackbonePlan, that plan's capacity-overflow word in the same synchronisation -- a stage
        that hit its active-site capacity has dropped rows, so the BEV map is wrong and the frame must not be returned.
        done (an event recorded behind the frame): waited for instead of the whole stream -- a later frame may already be queued there."""
        if not n_out.is_cuda:  # the pinned pair the last kernel wrote itself (native_proposals host_out): wait for the frame, read
            if done is not None:
                done.synchronize()
            else:
                torch.cuda.current_stream(boxes.device).synchronize()
            n, ovf = int(n_out[0]), int(n_out[1])
            _raise_on_flag(ovf)
        elif overflow_flag is None and n_out.numel() == 1:
            n = int(n_out.item())
        else:
            host = _pinned_pair(n_out.device)
            if n_out.numel() == 2:  # the proposal stage already put the plan's verdict next to the count: one copy
                host.copy_(n_out, non_blocking=True)
            else:
                host[0:1].copy_(n_out, non_blocking=True)
                host[1:2].copy_(overflow_flag, non_blocking=True)
            torch.cuda.current_stream(n_out.device).synchronize()
            n, ovf = int(host[0]), int(host[1])
            _raise_on_flag(ovf)
        return [boxes[:n], batch_idx[:n], class_idx[:n], scores[:n]]

    def inference_native(self, head_maps, anchors, overflow_flag=None):
        if not self.native_supported(head_maps.shape[0], anchors.numel() // (7 * self.cfg.NUM_CLASSES)):  # too large: op-by-op
            cls_map, reg_map = self.maps_from_fused(head_maps)
            out = self.inference_from_maps(cls_map, reg_map, anchors)
            if overflow_flag is not None:
                _raise_on_flag(int(overflow_flag.item()))
            return out
        return self.finalize_native(*self.native_proposals(head_maps, anchors, overflow_flag))

    def finalize(self, boxes, batch_idx, class_idx, scores, keep, n_keep):
        keep = keep[: int(n_keep.item())]
        boxes, batch_idx, class_idx, scores = (x[keep] for x in (boxes, batch_idx, class_idx, scores))
        mask = self._above_score_thresh(scores, class_idx)
        return [x[mask] for x in (boxes, batch_idx, class_idx, scores)]

    def inference_from_maps(self, cls_map, reg_map, anchors):
        score_map = cls_map.sigmoid_()
        B, n_cls = score_map.shape[:2]
        scores, anchor_idx = score_map.view(B, n_cls, -1).topk(self.TOPK, -1)
        boxes = self._decode(reg_map, anchors, anchor_idx)
        return self._multiclass_batch_nms(boxes, scores)

    def reshape_cls(self, cls_map):
        B, _, ny, nx = cls_map.shape
        return cls_map.view(B, self.cfg.NUM_CLASSES, self.cfg.NUM_YAW, ny, nx)

    def reshape_reg(self, reg_map):
        B, _, ny, nx = reg_map.shape
        return reg_map.view(B, self.cfg.NUM_CLASSES, self.cfg.BOX_DOF, -1, ny, nx).permute(0, 1, 3, 4, 5, 2)

    def forward(self, feature_map):
        if not self.training and not torch.is_grad_enabled() and feature_map.is_cuda:
            return self.maps_from_fused(self.native_head(feature_map))  # inference on the GPU: csrc/dense_conv.hip, not MIOpen
        return self.reshape_cls(self.conv_cls(feature_map)), self.reshape_reg(self.conv_reg(feature_map))

    def native_head(self, feature_map):
        """fp32 (B, C_IN, H, W) cuda -> fused [cls | reg] maps (B, n_anchor * (1 + DOF), H, W) fp32 on the streaming split-precision
        MFMA 1x1 kernel (csrc/dense_conv.hip:conv1x1_bf16x3_small_cout_kernel) in f16s arithmetic -- the input's scale entry is
        taken from the tensor's own maximum per call, so nothing can leave the range --; the packed weight image is cached until a
        head tensor changes.  (Second's inference paths get the same maps from DenseHeadPlan, which feeds the kernel split planes
        directly; this entry is for callers that hold an fp32 feature map: PV_RCNN.proposal, `model.head(features)`.)"""
        from ..runtime import conv2d_split, pack_conv_weight, to_split_nhwc
        tensors = (self.conv_cls.weight, self.conv_cls.bias, self.conv_reg.weight, self.conv_reg.bias)
        stamp = tuple((t.data_ptr(), t._version) for t in tensors)
        cache = self.__dict__.get("_native_head")
        if cache is None or cache[0] != stamp:
            with torch.no_grad():
                w = torch.cat((self.conv_cls.weight, self.conv_reg.weight), 0)
                bias = torch.cat((self.conv_cls.bias, self.conv_reg.bias), 0).float().contiguous()
                cache = (stamp, pack_conv_weight(w, None, "fp32"), bias, w.shape[1], w.shape[0])
            self.__dict__["_native_head"] = cache
        _, img, bias, cin, cout = cache
        hi, lo = to_split_nhwc(feature_map.float(), "fp32")
        return conv2d_split(hi, lo, img, bias, False, cin, cout, 1, out_split=False, out_nchw=True, pr=(hi.v3d_entry, None, None))[1]


class FusedProposalLossFunction(torch.autograd.Function):
    """ProposalLoss.forward and its gradient with respect to the fused head maps in one native pass (csrc/proposal_loss.hip):
    (maps, G_cls int8, M_cls, G_reg, M_reg) -> (cls_loss, reg_loss).  The gradient is computed with the forward; backward scales
    its two channel groups with the upstream gradients."""

    @staticmethod
    def forward(ctx, maps, g_cls, m_cls, g_reg, m_reg, n_cls, n_yaw, alpha, gamma):
        from .. import _lib as L
        b, _, h, w = maps.shape
        losses = torch.empty(3, dtype=torch.float32, device=maps.device)
        dmaps = torch.empty_like(maps)
        lib = L.lib()
        ws = L.workspace(lib.v3d_proposal_loss_workspace(), maps.device)
        with L.device_guard(maps.device):
            L.check(lib.v3d_proposal_loss_fwd_bwd(L.ptr(maps), L.ptr(g_cls), L.ptr(m_cls), L.ptr(g_reg), L.ptr(m_reg), b, n_cls, n_yaw, h, w,
                                                  float(alpha), float(gamma), L.ptr(losses), L.ptr(dmaps), L.ptr(ws), ws.numel(),
                                                  L.stream_ptr()), "proposal_loss_fwd_bwd")
        ctx.dmaps, ctx.geom = dmaps, (b, n_cls, n_yaw, h, w)
        return losses[0], losses[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_cls_loss, g_reg_loss):
        from .. import _lib as L
        dmaps, ctx.dmaps = ctx.dmaps, None
        if dmaps is None:
            raise RuntimeError("fused proposal loss: backward called twice (the gradient buffer is consumed by the first call)")
        gc = g_cls_loss.to(torch.float32).contiguous()
        gr = g_reg_loss.to(torch.float32).contiguous()
        with L.device_guard(dmaps.device):
            L.check(L.lib().v3d_proposal_loss_scale(L.ptr(dmaps), *ctx.geom, L.ptr(gc), L.ptr(gr), L.stream_ptr()), "proposal_loss_scale")
        return (dmaps,) + (None,) * 8


class ProposalLoss(nn.Module):
    """Focal classification + smooth-L1 box loss, both divided by max(#positives, 1)
    (proposal.py:100-141).  (P, G, M) = (predicted, ground truth, mask)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    @staticmethod
    def masked_sum(loss, mask):
        return (loss * mask.type_as(loss)).sum()

    def reg_loss(self, P_reg, G_reg, M_reg):
        per = F.smooth_l1_loss(P_reg, G_reg, reduction="none")
        total = per[..., 0:3] + per[..., 3:6] + per[..., 6:7] / math.pi
        return self.masked_sum(total, M_reg)

    def cls_loss(self, P_cls, G_cls, M_cls):
        return self.masked_sum(sigmoid_focal_loss(P_cls, G_cls.float(), reduction="none"), M_cls)

    def _fused(self, item):
        """The native pass applies when the model left its FUSED head maps in the item (`_head_maps`: Second.forward on the native
        training path) and the targets have the assigner's layout; else None and the torch expressions below run."""
        maps = item.get("_head_maps")
        if isinstance(maps, tuple):
            # (maps, P_cls, P_reg) as Second.forward leaves them: the fused maps only speak for this item while its P_cls / P_reg
            # are still the views made from them -- outputs that were replaced or post-processed go through the torch expressions
            maps, p_cls, p_reg = maps
            if item.get("P_cls") is not p_cls or item.get("P_reg") is not p_reg:
                return None
        cfg = self.cfg
        if maps is None or not maps.is_cuda or maps.dtype != torch.float32 or not maps.is_contiguous() or cfg.BOX_DOF != 7:
            return None
        G_cls, M_cls, G_reg, M_reg = (item[k] for k in ("G_cls", "M_cls", "G_reg", "M_reg"))
        b, o, h, w = maps.shape
        n_cls, n_yaw = cfg.NUM_CLASSES, cfg.NUM_YAW
        shape = (b, n_cls, n_yaw, h, w)
        if o != n_cls * n_yaw * 8 or tuple(G_cls.shape) != shape or tuple(M_cls.shape) != shape \
                or tuple(G_reg.shape) != shape + (7,) or tuple(M_reg.shape) != shape + (1,) or G_reg.dtype != torch.float32:
            return None
        if any(t.device != maps.device for t in (G_cls, M_cls, G_reg, M_reg)):
            return None
        g_cls = G_cls if G_cls.dtype == torch.int8 else G_cls.to(torch.int8)
        as_u8 = lambda m: (m if m.dtype in (torch.bool, torch.uint8) else m.ne(0)).contiguous().view(torch.uint8)
        cls_loss, reg_loss = FusedProposalLossFunction.apply(maps, g_cls.contiguous(), as_u8(M_cls), G_reg.contiguous(), as_u8(M_reg),
                                                             n_cls, n_yaw, 0.25, 2.0)
        return dict(cls_loss=cls_loss, reg_loss=reg_loss, loss=cls_loss + self.cfg.TRAIN.LAMBDA * reg_loss)

    def forward(self, item):
        fused = self._fused(item)
        if fused is not None:
            return fused
        G_cls, M_cls, P_cls, G_reg, M_reg, P_reg = (item[k] for k in ("G_cls", "M_cls", "P_cls", "G_reg", "M_reg", "P_reg"))
        normalizer = M_reg.type_as(P_reg).sum().clamp_(min=1)
        cls_loss = self.cls_loss(P_cls, G_cls, M_cls) / normalizer
        reg_loss = self.reg_loss(P_reg, G_reg, M_reg) / normalizer
        return dict(cls_loss=cls_loss, reg_loss=reg_loss, loss=cls_loss + self.cfg.TRAIN.LAMBDA * reg_loss)
