"""LMM front-end mirroring pyseer/lmm.py: fit_lmm_block / fit_lmm over the HIP engine."""
import numpy as np

from .classes import FLAG_FILTER, FLAG_PREFILTER, notes_from_flags

NOTE_LRT = 1 << 8


def mask_like_fit_lmm(r):
    """Apply the NaN masking the reference's fit_lmm leaves in its LMM tuples (pyseer/lmm.py:160-217):
    af/pre-filtered variants keep only prep; LRT-filtered ones keep prep and pvalue."""
    fl = r["flags"]
    pre = (fl & FLAG_PREFILTER) != 0
    af = (fl & 1) != 0
    lrt = (fl & NOTE_LRT) != 0
    out = dict(r)
    for f in ("pvalue", "beta", "bse", "frac_h2"):
        a = np.array(r[f], dtype=float, copy=True)
        a[pre] = np.nan
        if f != "pvalue":
            a[lrt] = np.nan
        out[f] = a
    p = np.array(r["prep"], dtype=float, copy=True)
    p[af] = np.nan
    out["prep"] = p
    return out
