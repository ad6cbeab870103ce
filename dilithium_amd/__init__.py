"""dilithium_amd -- MI355X (gfx950) NTT hot path for CRYSTALS-Dilithium, behind the
GMUCERG/Dilithium dilithium-256/ C++ API.

The product is libdil256.so (hand-written HIP kernels + an extern "C" boundary, see
include/dil256.h).  This package is the thin Python host side used by tests and bench.py:
  lib       ctypes loader of the C-ABI (fails loudly when the library or the GPU is missing)
  api       mirror of the reference's function names on torch device tensors / numpy arrays
  sharding  one-process-per-GPU batch sharding + the final RCCL gather
"""
from .lib import DilError, load  # noqa: F401

Q = 8380417
N = 256
NATURAL, AFTER_NTT, AFTER_INVNTT = 0, 1, 2
LEVELS = {2: (4, 4), 3: (6, 5), 5: (8, 7)}
