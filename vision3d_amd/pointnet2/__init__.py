"""The slice of sshaoshuai/Pointnet2.PyTorch that vision3d calls (detector/model.py:6-7,
detector/roi_grid_pool.py:5), on MI355X.  The upstream package is absent from the reference tree:
semantics follow the published kernels as restated in oracle/v3d_oracle.c ("parity unpinned")."""
from . import pointnet2_utils, pointnet2_modules  # noqa: F401
