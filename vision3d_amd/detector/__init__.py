"""`vision3d.detector` surface (vision3d/detector/__init__.py:1-3)."""
from .model import PV_RCNN
from .second import Second
from .proposal import ProposalLoss, ProposalLayer
