"""`vision3d.ops` surface (vision3d/ops/__init__.py:1-4), served by libvision3d_hip.so."""
from .matcher import Matcher, subsample_labels
from .focal_loss import sigmoid_focal_loss
from .iou_nms import batched_nms, batched_nms_rotated, nms, nms_rotated, box_iou_rotated
from .iou_nms import batched_nms_rotated_padded, nms_rotated_padded
