"""`vision3d.core` surface that sits on the hot path (vision3d/core/__init__.py:1-5; the visdom plotter
and the dead refinement-target assigner are out of scope)."""
from .config import cfg
from .anchor_generator import AnchorGenerator
from .preprocess import TrainPreprocessor, Preprocessor
from .proposal_targets import ProposalTargetAssigner
