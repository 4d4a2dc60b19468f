"""numpy/BLAS restatement of the reference's block LMM test, used ONLY as bench.py's cpu_baseline leg and by tests.

TEST INFRASTRUCTURE: nothing under pyseer_amd/ may import this module.  It follows the reference line by line where the C
oracle (seer_oracle.c) loops: the rotation is the dense `U.T.dot(A)` the reference itself issues through BLAS
(pyseer/fastlmm/lmm_cov.py:186), so its speed per core is the reference's, not that of a scalar port.

  pyseer/lmm.py:228-260                 fit_lmm_block
  pyseer/fastlmm/lmm_cov.py:165-194     rotate            (Linreg.regress :874-880: residual on the covariates)
  pyseer/fastlmm/lmm_cov.py:597-838     nLLeval / nLLcore (Sd, AKA, AKB, beta, variance, fraction explained)
"""
import numpy as np


class LmmBlas(object):
    """U (n,k), S (k,), y (n,), covar (n,D) with the intercept LAST (lmm.py:95-99)."""

    def __init__(self, U, S, y, covar):
        self.U = np.ascontiguousarray(U, dtype=np.float64)
        self.S = np.asarray(S, dtype=np.float64)
        self.n, self.k = self.U.shape
        self.C = np.asarray(covar, dtype=np.float64).reshape(self.n, -1)
        self.D = self.C.shape[1]
        self.Cd = np.linalg.pinv(self.C)                                   # Linreg.Xdagger, lmm_cov.py:869
        self.UY = self._rotate(np.asarray(y, dtype=np.float64).reshape(self.n, 1))[:, 0]

    def _rotate(self, A):
        """lmm_cov.py:165-194: residualise on the covariates, zero columns they explain, rotate by U^T."""
        A = A - self.C.dot(self.Cd.dot(A))                                 # regress(), :874-880
        A[:, A.std(0) <= 1e-10] = 0.0                                      # :179-181
        return self.U.T.dot(A)                                             # :186  (the 2 k N flop per variant)

    def block(self, h2, Xv):
        """Xv: (B, n) variant-major 0/1.  Returns (beta, bse, frac_h2, p) as fit_lmm_block does."""
        from scipy import stats
        if not (0.0 <= h2 < 1.0):
            raise KeyError("beta")                                         # lmm_cov.py:667-670
        Sd = h2 * self.S + (1.0 - h2)                                      # :665
        UX = self._rotate(np.ascontiguousarray(np.asarray(Xv, dtype=np.float64).T))
        yKy = np.sum(self.UY * self.UY / Sd)
        W = UX / Sd[:, None]
        sKs = np.einsum("ij,ij->j", UX, W)                                 # computeAKA, :885-900
        sKy = W.T.dot(self.UY)                                             # computeAKB, :902-916
        with np.errstate(divide="ignore", invalid="ignore"):
            beta = sKy / sKs                                               # :802
            beta[np.isnan(beta) & (sKy == 0.0)] = 0.0                      # :803-805
            veb = sKy * beta
            var = (yKy - veb) / ((self.n - self.D) - 1.0) / sKs            # :813
            frac = np.sqrt(veb / yKy)                                      # :814, lmm.py:258
            chi2 = beta * beta / var                                       # lmm.py:248
            p = stats.f.sf(chi2, 1, self.n - (self.D + 1))                 # lmm.py:251-253
        return beta, np.sqrt(var), frac, p
