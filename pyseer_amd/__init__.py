"""pyseer_amd -- MI355X-native per-variant association engine behind pyseer's interface.

Only the per-variant hot path of pyseer lives here (prefilter, fixed-effects logistic/Firth/OLS,
FaST-LMM test), as hand-written HIP kernels for gfx950 in csrc/, reached through the flat C ABI
of libseerhip.so (include/seerhip.h) via ctypes (_abi.py).  There is no CPU fallback.
"""
__version__ = "0.1.0"

from .classes import Seer, LMM  # noqa: F401
