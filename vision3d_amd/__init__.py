"""vision3d_amd -- the vision3d point-cloud hot path (voxelizer, sparse 3-D conv backbone, rotated
IoU/NMS, points-in-boxes, PV-RCNN point ops) as hand-written HIP for MI355X (gfx950) behind the
reference's own Python surface:

    vision3d.ops        -> vision3d_amd.ops         vision3d._C      -> vision3d_amd._C
    vision3d.core       -> vision3d_amd.core        vision3d.detector -> vision3d_amd.detector
    spconv (subset)     -> vision3d_amd.spconv      pointnet2 (subset) -> vision3d_amd.pointnet2

Native code: vision3d_amd/lib/libvision3d_hip.so (C ABI: include/vision3d_hip.h).  No CPU fallback.
"""
__version__ = "0.1"
