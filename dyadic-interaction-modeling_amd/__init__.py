"""dimx -- MI355X-native DIM-Listener hot path (speaker motion + audio -> listener motion).

Host side: Python on PyTorch-ROCm mirroring the reference's model/config surface.
All arithmetic of the path runs in hand-written HIP kernels for gfx950 behind the C-ABI
declared in ``include/dimx.h`` (``csrc/`` -> ``libdimx_hip.so``, loaded with ctypes).
"""
from . import config, prng, weights  # noqa: F401

__version__ = "0.1.0"
