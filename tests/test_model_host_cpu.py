"""Host-side null fits (pyseer_amd/model.py fit_null) against the reference's numbers (model_test.py:120-172 family)."""
import os

import pytest
import numpy as np

from pyseer_amd.model import fit_null, fit_firth_host, firth_likelihood

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fit_null_matches_reference():
    d = np.load(os.path.join(G, "model_unit.npz"))
    pb, pc, m, cov = d["p_binary"], d["p_continuous"], d["m"], d["cov"]
    e = np.zeros((0, 0))
    assert abs(fit_null(pb, m, e, False).llf - float(d["null_llf_binary"])) < 1e-9
    assert abs(fit_null(pb, m, e, False, firth=True) - float(d["null_firth_binary"])) < 1e-8
    assert abs(fit_null(pc, m, e, True).llf - float(d["null_llf_cont"])) < 1e-8
    assert abs(fit_null(pb, m, cov, False).llf - float(d["null_llf_binary_cov"])) < 1e-9
    assert abs(fit_null(pb, m, cov, False, firth=True) - float(d["null_firth_binary_cov"])) < 1e-8


def test_fit_firth_host_matches_reference():
    d = np.load(os.path.join(G, "model_unit.npz"))
    pb, k = d["p_binary"], d["k"]
    X = np.concatenate((np.ones((100, 1)), k.reshape(-1, 1)), axis=1)
    sv = np.array([np.log(pb.mean() / (1 - pb.mean())), 0.0])
    beta, bse, fitll = fit_firth_host(X, pb, sv)
    want = d["fit_firth_k"]
    assert np.allclose([beta[0], beta[1], bse, fitll], [want[0], want[1], want[3], want[4]], rtol=1e-8, atol=1e-9)


def test_synthetic_nulls():
    for name in ("glm_N300_q10", "glm_N100_q0", "glm_N1000_q10"):
        d = np.load(os.path.join(G, name + ".npz"))
        m = d["m"] if int(d["q"]) else np.zeros((0, 0))
        assert abs(fit_null(d["y"], m, np.zeros((0, 0)), False).llf - float(d["null_llf"])) < 1e-8
        assert abs(fit_null(d["y"], m, np.zeros((0, 0)), False, firth=True) - float(d["null_firth"])) < 1e-7


def test_fit_null_degenerate_designs_match_reference():
    """Null fits on rank-deficient designs (tests/golden/make_null_golden.py ran the reference): the ridge keeps Newton going
    on a duplicated column; when the final un-ridged inverse fails here, the Powell fallback (model.py:134-137) lands on the
    same log-likelihood."""
    from pyseer_amd.model import fit_null
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "null_degenerate.npz"))
    p, m = d["p"], d["m"]
    n = p.shape[0]
    e = np.zeros((n, 0))
    for name, mm in (("dup", np.c_[m, m[:, 0]]), ("const", np.c_[m, np.ones(n)]), ("ok", m)):
        r = fit_null(p, mm, e, False)
        assert r is not None and abs(r.llf - float(d["llf_" + name])) < 1e-8, name
    assert fit_null(p, m, e, False, firth=True) == pytest.approx(float(d["firth_ok"]), rel=1e-9)
    assert fit_null(p, np.c_[m, np.ones(n)], e, False, firth=True) == -np.inf


def test_missing_calls_follow_the_reference_order_of_checks():
    """A variant with missing calls (NaN in k) through the by-name drop-in: prefilter first, then the MissingDataError return
    (pyseer/model.py:262-271, 371-377).  Expected values recorded from the reference (tests/golden/make_missing_golden.py).  No GPU:
    such variants never reach the engine."""
    import json
    import os
    import pandas as pd
    from pyseer_amd.model import fixed_effects_regression
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "missing_data.json")))
    assert len(d["cases"]) == 8
    for c in d["cases"]:
        p = np.array(c["p"]); k = np.array([np.nan if x is None else x for x in c["k"]]); m = np.array(c["m"])
        cov = pd.DataFrame(np.zeros((len(p), 0)))
        s = fixed_effects_regression("v", p, k, m, cov, 0.4, b"x", False, None, c["pret"], 1.0, None, None, [], [], c["continuous"])
        assert sorted(s.notes) == c["notes"] and s.prefilter == c["prefilter"] and s.filter == c["filter"]
        assert abs(s.prep - c["prep"]) <= 1e-12 * abs(c["prep"])
        assert np.isnan(s.pvalue) and np.isnan(s.kbeta) and s.max_lineage is None
