"""The `vision3d.ops` names (reference: vision3d/ops/__init__.py), served by libvision3d_hip.so, plus the two
sync-free NMS forms used inside captured graphs.  Resolved on first access."""
import importlib

_EXPORTS = {
    "Matcher": "matcher", "subsample_labels": "matcher",
    "sigmoid_focal_loss": "focal_loss",
    "box_iou_rotated": "iou_nms", "box_iou_rotated_3d": "iou_nms", "nms": "iou_nms", "nms_rotated": "iou_nms",
    "batched_nms": "iou_nms", "batched_nms_rotated": "iou_nms",
    "nms_rotated_padded": "iou_nms", "batched_nms_rotated_padded": "iou_nms",
}
__all__ = sorted(_EXPORTS)


def __getattr__(name):
    try:
        module = importlib.import_module("." + _EXPORTS[name], __name__)
    except KeyError:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}") from None
    value = getattr(module, name)
    globals()[name] = value
    return value


def __dir__():
    return sorted(list(globals()) + __all__)
