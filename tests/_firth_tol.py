"""Tolerance for Firth-fitted rows, as the round-2 review asked for it: 1e-6 relative, and an absolute slack (the halving test's noise floor,
half of fit_firth's last step) ONLY on rows the tie detector identifies -- rows on which the reference's own answer moves when the first
comparison of each iteration is biased by +-2e-13 |F| or steps below 1e-10 are accepted (orc.firth_noise_variants: the answers the
reference can legitimately give when `F(new) > F(old)` sits on the last bits of F).  Test infrastructure."""
import numpy as np

N_TIE_VARIANTS = 6  # oracle.FIRTH_NOISE_VARIANTS[:6] = the halving-tie settings, [6:] = the stop-rule settings (conv_scale)
FA = 3e-7          # noise floor of the halving test: half of a last step of ~6e-7 (case 1)
TINY = 1e-12       # |kbeta| itself can be ~1e-6: below this the relative test has no meaning


def firth_rows_close(got, variants, field, rows, rtol=1e-6):
    """got: array over all rows; variants: the list of oracle outputs (dicts of arrays), variants[0] = the reference's behaviour exactly;
    rows: boolean mask of the Firth-fitted rows to hold.  Returns (ok mask over rows, number of rows that needed the tie detector)."""
    g = np.asarray(got, dtype=float)[rows]
    base = np.asarray(variants[0][field], dtype=float)[rows]
    with np.errstate(invalid="ignore"):
        same = (np.isnan(g) & np.isnan(base)) | (np.abs(g - base) <= rtol * np.abs(base) + TINY)
        sens = np.zeros(g.shape, dtype=bool)                          # the reference's own answer depends on a last-bit comparison
        near = np.zeros(g.shape, dtype=bool)
        for v in variants[1:]:
            w = np.asarray(v[field], dtype=float)[rows]
            sens |= ~((np.isnan(w) & np.isnan(base)) | (np.abs(w - base) <= 1e-9 * np.abs(base) + 1e-13))
        for v in variants:
            w = np.asarray(v[field], dtype=float)[rows]
            near |= (np.isnan(g) & np.isnan(w)) | (np.abs(g - w) <= rtol * np.abs(w) + FA)
    ok = same | (sens & near)
    return ok, int((~same & sens & near).sum())


def golden_firth_rows_close(got, golden, variants, field, rows, rtol=1e-6):
    """The same rule against numbers the REFERENCE produced: `golden` over all rows, `variants` the oracle's answers under the tie detector's
    settings (only used to tell whether the row's halving test sits on rounding noise), `rows` the Firth-routed rows to hold.  A row passes at
    rtol with no slack, or -- only if the detector fires on it -- at rtol + FA.  Returns (ok mask over rows, rows that needed the detector)."""
    g = np.asarray(got, dtype=float)[rows]; w = np.asarray(golden, dtype=float)[rows]
    base = np.asarray(variants[0][field], dtype=float)[rows]
    with np.errstate(invalid="ignore"):
        same = (np.isnan(g) & np.isnan(w)) | (np.abs(g - w) <= rtol * np.abs(w) + TINY)
        differs = lambda x: ~((np.isnan(x) & np.isnan(base)) | (np.abs(x - base) <= 1e-9 * np.abs(base) + 1e-13))
        # the halving-tie settings (tie bias, accept-below): slack FA on rows where one of them moves the oracle's own answer
        sens = np.zeros(g.shape, dtype=bool)
        for v in variants[1:N_TIE_VARIANTS]:
            sens |= differs(np.asarray(v[field], dtype=float)[rows])
        near = (np.isnan(g) & np.isnan(w)) | (np.abs(g - w) <= rtol * np.abs(w) + FA)
        # the stop-rule settings (conv_scale 1 +- 1e-4: a stop met within 1e-8 of the limit): on a slowly (linearly) converging fit that moves
        # the answer by a whole step (~1e-4) -- such a row must then EQUAL, at rtol and without slack, the oracle's answer under the setting
        # that moved it.  (Round 4 let `alt` match any variant)
        alt = np.zeros(g.shape, dtype=bool)
        for v in variants[N_TIE_VARIANTS:]:
            x = np.asarray(v[field], dtype=float)[rows]
            alt |= differs(x) & (np.abs(g - x) <= rtol * np.abs(x) + TINY)
    ok = same | (sens & near) | alt
    return ok, int((~same & ok).sum())
