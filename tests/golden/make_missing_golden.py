"""fixed_effects_regression on variants that carry missing calls (NaN in k), recorded from the reference (pyseer/model.py:202-394):
the prefilter runs first (NaN falls in no cell of the 2x2 table), then statsmodels raises MissingDataError.
Run: PYTHONPATH=_harness:_harness/stubs:/root/reference /opt/conda/bin/python3.9 -W ignore make_missing_golden.py"""
import shim  # noqa: F401
import json
import numpy as np
import pandas as pd
from pyseer.model import fixed_effects_regression, fit_null

rng = np.random.default_rng(11)
n = 150
m = rng.standard_normal((n, 2))
cov = pd.DataFrame(np.zeros((n, 0)))
out = {"cases": []}
for cont in (False, True):
    p = (m[:, 0] + rng.standard_normal(n)) if cont else (rng.random(n) < 1 / (1 + np.exp(-(0.2 + m[:, 0])))).astype(float)
    null = fit_null(p, m, cov, cont)
    nf = None if cont else fit_null(p, m, cov, False, firth=True)
    for af, pret in ((0.4, 1.0), (0.4, 1e-6), (0.03, 1.0), (0.6, 0.5)):
        k = (rng.random(n) < af).astype(float)
        k[rng.choice(n, 5, replace=False)] = np.nan
        s = fixed_effects_regression("v", p, k, m, cov, af, b"x", False, None, pret, 1.0, null, nf, [], [], cont)
        out["cases"].append(dict(continuous=cont, pret=pret, p=p.tolist(), k=[None if np.isnan(x) else x for x in k], m=m.tolist(),
                                 prep=None if not np.isfinite(s.prep) else float(s.prep), notes=sorted(s.notes),
                                 prefilter=bool(s.prefilter), filter=bool(s.filter),
                                 pvalue_nan=bool(np.isnan(s.pvalue)), kbeta_nan=bool(np.isnan(s.kbeta))))
json.dump(out, open(__file__.replace("make_missing_golden.py", "missing_data.json"), "w"))
print([(c["continuous"], c["pret"], c["prep"], c["notes"], c["prefilter"], c["filter"]) for c in out["cases"]])
