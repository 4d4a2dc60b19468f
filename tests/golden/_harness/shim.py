"""Oracle-harness shim (runs ONLY in the build container, never on the GPU box).

statsmodels 0.12.2 reads np.MachAr at import; numpy 1.26 removed it. Install a
stand-in BEFORE statsmodels is imported. Touches neither the reference nor conda.
"""
import numpy as np

if not hasattr(np, "MachAr"):
    class _MachAr(object):
        def __init__(self, *a, **k):
            f = np.finfo(float)
            self.eps = f.eps
            self.tiny = f.tiny
            self.huge = f.max
            self.epsneg = f.epsneg
            self.precision = f.precision
            self.resolution = f.resolution
    np.MachAr = _MachAr
