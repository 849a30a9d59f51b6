"""Detector-side names a vision3d user imports (`PV_RCNN`, `Second`, `ProposalLoss`; reference:
vision3d/detector/__init__.py).  Resolved on first access so that `import vision3d_amd.detector` stays cheap
and a missing GPU library only surfaces when a model is actually built."""
import importlib

_EXPORTS = {
    "PV_RCNN": "model",
    "Second": "second",
    "ProposalLoss": "proposal",
    "ProposalLayer": "proposal",
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
