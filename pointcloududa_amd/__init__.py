"""pointcloududa_amd -- MI355X-native (gfx950) implementation of the PointCloudUDA adversarial
train-step hot path: hand-written HIP kernels (libpcuda_hip.so, C ABI in include/pcuda_hip.h)
behind the reference's ``src/networks`` module surface.  There is no CPU or ATen fallback."""
from . import _lib  # noqa: F401
from .kernels import get_precision, set_precision  # noqa: F401

__all__ = ["set_precision", "get_precision"]
