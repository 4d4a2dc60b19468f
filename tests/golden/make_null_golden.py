"""Null-model fits on degenerate designs, recorded from the reference (pyseer.model.fit_null).
Run: PYTHONPATH=_harness:_harness/stubs:/root/reference /opt/conda/bin/python3.9 -W ignore make_null_golden.py"""
import shim  # noqa: F401
import numpy as np
import pandas as pd
from pyseer.model import fit_null

rng = np.random.default_rng(3)
n = 120
m = rng.standard_normal((n, 3))
p = (rng.random(n) < 1 / (1 + np.exp(-(0.3 + m[:, 0])))).astype(float)
cov = pd.DataFrame(np.zeros((n, 0)))
out = dict(p=p, m=m)
for name, mm in (("dup", np.c_[m, m[:, 0]]), ("const", np.c_[m, np.ones(n)]), ("ok", m)):
    r = fit_null(p, mm, cov, False)
    out["llf_" + name] = np.nan if r is None else float(r.llf)
    out["firth_" + name] = np.nan
    f = fit_null(p, mm, cov, False, firth=True)
    if f is not None:
        out["firth_" + name] = float(f)
np.savez(__file__.replace("make_null_golden.py", "null_degenerate.npz"), **out)
print({k: v for k, v in out.items() if k not in ("p", "m")})
