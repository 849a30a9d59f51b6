"""Training-time augmentation on the device (reference: vision3d/dataset/augmentation.py).
The KITTI readers and the database builder (rank 4) are not part of this package."""
from .augmentation import (ChainedAugmentation, FlipAugmentation, RotateAugmentation, SampleAugmentation,  # noqa: F401
                           SampleDatabase, ScaleAugmentation)
