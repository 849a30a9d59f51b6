"""Anchor <-> ground-truth matching (interface of vision3d/core/proposal_targets.py:10-88).

`forward` runs the fused device kernels of csrc/targets.hip (IoU, column/row maxima, band labels, low-quality
rule and box encoding in two launches, no n_gt x 70 400 matrix); `forward_torch` is the op-by-op statement of the
reference (IoU matrix in csrc/iou_nms.hip + torch Matcher/encode) kept as the on-device cross-check.
Emits G_cls/M_cls (n_cls, n_yaw, ny, nx), G_reg (..., 7), M_reg (..., 1).
"""
import torch
from torch import nn

from .. import _lib as L
from ..ops import Matcher, box_iou_rotated
from .anchor_generator import AnchorGenerator
from .box_encode import encode

BEV = [0, 1, 3, 4, 6]  # (x, y, w, l, yaw): yaw stays in radians although the op reads degrees (H1)


class ProposalTargetAssigner(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.anchors = AnchorGenerator(cfg).anchors.cuda()
        self.matchers = [Matcher(a["iou_thresh"], [0, -1, +1], cfg.ALLOW_LOW_QUALITY_MATCHES) for a in cfg.ANCHORS]

    def compute_iou(self, boxes, anchors):
        return box_iou_rotated(boxes[:, BEV].contiguous(), anchors[:, BEV].contiguous())

    def match_class_i(self, boxes, class_idx, full_idx, i):
        sel = class_idx == i
        anchors = self.anchors[i].reshape(-1, self.cfg.BOX_DOF)
        matches, labels = self.matchers[i](self.compute_iou(boxes[sel], anchors))
        if sel.any():
            matches = full_idx[sel][matches]
        return matches, labels

    def apply_ignore_mask(self, matches, labels, box_ignore):
        labels[box_ignore[matches] & (labels != -1)] = -1

    def match_all_classes(self, boxes, class_idx, box_ignore):
        full_idx = torch.arange(boxes.shape[0], device=boxes.device)
        pairs = [self.match_class_i(boxes, class_idx, full_idx, i) for i in range(self.cfg.NUM_CLASSES)]
        shape = self.anchors.shape[:-1]
        matches = torch.stack([p[0] for p in pairs]).view(shape)
        labels = torch.stack([p[1] for p in pairs]).view(shape)
        return matches, labels

    def get_cls_targets(self, G_cls):
        return G_cls.clamp(min=0), G_cls.ne(-1)

    def get_reg_targets(self, boxes, box_idx, G_cls):
        pos = G_cls == 1
        G_reg = torch.zeros_like(self.anchors)
        G_reg[pos] = encode(boxes[box_idx[pos]], self.anchors[pos])
        return G_reg, pos.unsqueeze(-1)

    def forward(self, item):
        """Fused path: two launches of csrc/targets.hip."""
        dev = self.anchors.device
        boxes = L.as_f32("assign_targets", item["boxes"].to(dev)).reshape(-1, self.cfg.BOX_DOF)
        class_idx = item["class_idx"].to(dev).to(torch.int64).contiguous()
        n_gt, n_cls = boxes.shape[0], self.cfg.NUM_CLASSES
        if self.cfg.BOX_DOF != 7 or n_cls > 16 or n_gt > 128:  # beyond the kernel's staging limits
            return self.forward_torch(item)
        A = self.anchors[0].numel() // 7
        shape = self.anchors.shape[:-1]
        G_cls = torch.empty(shape, dtype=torch.int8, device=dev)
        M_cls = torch.empty(shape, dtype=torch.bool, device=dev)
        G_reg = torch.empty(self.anchors.shape, dtype=torch.float32, device=dev)
        M_reg = torch.empty(shape + (1,), dtype=torch.bool, device=dev)
        lib = L.lib()
        ws = L.workspace(lib.v3d_assign_targets_workspace(n_gt, n_cls, A), dev)
        thresh = L.host_f32([t for a in self.cfg.ANCHORS[:n_cls] for t in a["iou_thresh"]])
        with torch.cuda.device(dev):
            L.check(lib.v3d_assign_targets(L.ptr(boxes), L.ptr(class_idx), n_gt, L.ptr(self.anchors), n_cls, A, thresh,
                                           int(bool(self.cfg.ALLOW_LOW_QUALITY_MATCHES)), L.ptr(G_cls), L.ptr(M_cls), L.ptr(G_reg),
                                           L.ptr(M_reg), 0, L.ptr(ws), ws.numel(), L.stream_ptr()), "assign_targets")
        item.update(G_cls=G_cls, G_reg=G_reg, M_cls=M_cls, M_reg=M_reg)
        return item

    def forward_torch(self, item):
        dev = self.anchors.device
        boxes, class_idx, box_ignore = (item[k].to(dev) for k in ("boxes", "class_idx", "box_ignore"))
        box_idx, G_cls = self.match_all_classes(boxes, class_idx, box_ignore)
        G_cls, M_cls = self.get_cls_targets(G_cls)
        G_reg, M_reg = self.get_reg_targets(boxes, box_idx, G_cls)
        item.update(G_cls=G_cls, G_reg=G_reg, M_cls=M_cls, M_reg=M_reg)
        return item
