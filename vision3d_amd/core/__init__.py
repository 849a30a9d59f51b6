"""The part of `vision3d.core` that sits on the hot path (reference: vision3d/core/__init__.py; its visdom plotter
and the unused refinement-target assigner are out of scope).  Resolved on first access."""
import importlib

_EXPORTS = {
    "cfg": "config",
    "AnchorGenerator": "anchor_generator",
    "Preprocessor": "preprocess", "TrainPreprocessor": "preprocess",
    "ProposalTargetAssigner": "proposal_targets",
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
