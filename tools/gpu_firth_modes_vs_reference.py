"""The three Firth step-halving modes against the REAL reference (pyseer.model.fit_firth, statsmodels 0.12.2) on the disputed rows of the
C4 workload: tests/golden/n5000_firth.npz holds the 82 rows on which the modes or the C restatement disagreed (firth-fail flag, or
|dkbeta| > 1e-7) plus 24 controls, with the reference's own answer for each (tests/golden/make_n5000_golden.py).
  noise     (default) an increase of F within 4 ulp is not an increase, steps below 1e-10 are accepted
  literal   SEERHIP_ROUTE firth_literal=1: `F(new) > F(old)` on the rounds' own evaluation of F
  strict    SEERHIP_ROUTE firth_strict=1: literal rule, one log per sample, the reference's start vector
Also the CPU restatement (oracle/seer_oracle.c) as it is now (log-likelihood summed in numpy's pairwise order) .
Writes gpurun_out/r04/firth_modes_vs_real_reference.json (copied to profiles/r04/)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyseer_amd.engine import Engine
from oracle import oracle as orc

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(root, "tests", "golden", "n5000_firth.npz"))
N, q = int(d["N"]), int(d["q"])
y = d["y"].astype(float); W = d["W"]; bits = np.ascontiguousarray(d["bits"]); fm = d["firth_main"]
ref_fit = d["firth_ok"] == 1


def score(kb, bse, ic, fail):
    both = ref_fit & ~fail
    dk = np.abs(kb[both] - fm[both, 1]); rel = dk / np.maximum(np.abs(fm[both, 1]), 1e-300)
    return {"firth_fail": int(fail.sum()), "rows_with_a_flag_difference": int((fail != ~ref_fit).sum()), "max_abs_dkbeta": float(dk.max()),
            "max_rel_dkbeta": float(rel.max()), "rows_rel_dkbeta_over_1e-6": int((rel > 1e-6).sum()), "rows_abs_dkbeta_over_1e-7": int((dk > 1e-7).sum()),
            "max_rel_dbse": float(np.max(np.abs(bse[both] - fm[both, 2]) / fm[both, 2])), "max_abs_dintercept": float(np.max(np.abs(ic[both] - fm[both, 0]))),
            "rows_right_at_1e-6": int((~(fail != ~ref_fit) & np.where(both, np.abs(kb - fm[:, 1]) <= 1e-6 * np.abs(fm[:, 1]), True)).sum())}


res = {"rows": int(bits.shape[0]), "disputed_rows": int(d["disputed"].sum()), "reference": {"fit": int(ref_fit.sum()), "firth_fail": int((~ref_fit).sum())},
       "what": "N=5000, q=10, force_firth; rows from the C4 workload (seed 4242, first 24576 rows) on which round 3's modes / restatement disagreed, + 24 controls; "
               "judge = pyseer.model.fit_firth itself (statsmodels 0.12.2, numpy 1.26.4)"}
w = orc.firth_batch(y, np.unpackbits(bits, axis=1, bitorder="little")[:, :N].astype(float), W)
res["restatement_pairwise_loglike"] = score(w["kbeta"], w["bse"], w["intercept"], w["status"] != 0)
res["restatement_round3_running_sum"] = score(d["r3_restatement_kbeta"], d["r3_restatement_bse"], d["r3_restatement_intercept"], d["r3_restatement_fail"])
for mode, env in (("noise", {}), ("literal", {"SEERHIP_ROUTE": "firth_literal=1"}), ("strict", {"SEERHIP_ROUTE": "firth_strict=1"})):
    for k, v_ in env.items():
        os.environ[k] = v_
    e = Engine(N); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, float(d["null_llf"]), float(d["null_firth"]), force_firth=True)
    r = e.glm_batch(bits); e.close()
    for k in env:
        del os.environ[k]
    res[mode] = score(r["kbeta"], r["bse"], r["intercept"], ((r["flags"] >> 6) & 1) == 1)
print(json.dumps(res, indent=1))
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", root), "gpurun_out", "r04"); os.makedirs(o, exist_ok=True)
json.dump(res, open(os.path.join(o, "firth_modes_vs_real_reference.json"), "w"), indent=1)
