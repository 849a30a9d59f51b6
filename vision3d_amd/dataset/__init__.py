"""Either side of the hot path in training: GT-sampling + global augmentation on the device
(rank 2; reference vision3d/dataset/augmentation.py) and the KITTI file readers (rank 4; reference
vision3d/dataset/kitti_utils.py, host numpy).  The dataset / annotation-cache classes are not part of this package.
Resolved on first access."""
import importlib

_EXPORTS = {
    "ChainedAugmentation": "augmentation", "SampleAugmentation": "augmentation", "FlipAugmentation": "augmentation",
    "ScaleAugmentation": "augmentation", "RotateAugmentation": "augmentation", "SampleDatabase": "augmentation",
    "read_points": "kitti", "read_labels": "kitti", "read_calib": "kitti", "boxes_in_lidar_frame": "kitti",
    "crop_to_camera_view": "kitti", "load_frame": "kitti", "Calib": "kitti", "Labels": "kitti",
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
