"""claymore_amd - MI355X-native MPM substep engine (the hot path of penn-graphics-research/claymore).

Product path = libclaymore_hip.so (hand-written gfx950 kernels) behind the C ABI of
include/claymore_amd.h; this package is the thin host-side mirror used by bench.py, the tests and
the multi-GPU driver.  There is no CPU fallback.
"""
from . import _ffi  # noqa: F401
from ._ffi import J_FLUID, FIXED_COROTATED, SAND, NACC  # noqa: F401

__all__ = ["_ffi", "engine", "scenes", "J_FLUID", "FIXED_COROTATED", "SAND", "NACC"]
