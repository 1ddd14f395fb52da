"""gslam_amd — MI355X (gfx950) hot path of GSLAM: ORB front end, brute-force Hamming matcher and
LM bundle adjustment as hand-written HIP kernels behind a plain-C ABI (include/gslam_hip.h).

The Python layer is a thin ctypes mirror of that ABI used by tests and bench.py; the product is
gslam_amd/lib/libgslam_hip.so plus the GSLAM plugin shims next to it.  There is no CPU fallback:
importing `gslam_amd.hip` without the built library raises, and every call needs a GPU.
"""
from . import hip  # noqa: F401

__all__ = ["hip"]
