"""Print the rows of one fuzz configuration on which the HIP path and the oracle disagree (debug aid for tests/test_fuzz_gpu.py).
usage: python tools/gpu_fuzz_case.py fixed|firth SEED"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_fuzz_gpu as T
from oracle import oracle as orc
from pyseer_amd.engine import Engine, pack_variants
from pyseer_amd.model import fit_null

kind, seed = sys.argv[1], int(sys.argv[2])
e0 = np.zeros((0, 0))
if kind == "fixed":
    N, q, cont, W, y, pret, lrtt, K = T._fixed_case(seed)
    null = fit_null(y, W, e0, cont); nf = np.nan if cont else fit_null(y, W, e0, False, firth=True)
    run = lambda: orc.fixed_effects_batch(y, K.astype(float), W if q else None, cont, pret, lrtt, null.llf, nf)
    e = Engine(N); e.glm_setup(y, W, cont, null.llf, nf, pret, lrtt)
    fields = ("prep", "pvalue", "kbeta", "bse", "intercept")
else:
    N, q, W, y, K = T._firth_case(seed); cont = False
    null = fit_null(y, W, e0, False); nf = fit_null(y, W, e0, False, firth=True)
    run = lambda: orc.firth_batch(y, K.astype(float), W if q else None)
    e = Engine(N); e.glm_setup(y, W, False, null.llf, nf, force_firth=True)
    fields = ("kbeta", "bse", "intercept")
r = e.glm_batch(pack_variants(K)); e.close()
alts = orc.firth_noise_variants(run)          # [reference, tie+, tie-, accept<1e-10, tie+ & accept, tie- & accept]
want = alts[0]
print("N", N, "q", q, "cont", cont, "V", len(K))
bad = np.zeros(len(K), bool)
for f in fields:
    with np.errstate(invalid="ignore"):
        bad |= ~(np.isclose(r[f], want[f], rtol=2e-6, atol=1e-6) | (np.isnan(r[f]) & np.isnan(want[f])))
if kind == "fixed":
    bad |= (r["flags"] & 0x1FF) != want["notes"]
else:
    bad |= (((r["flags"] >> 6) & 1) == 1) != (want["status"] != 0)
for i in np.flatnonzero(bad):
    print("row", i, "carriers", int(K[i].sum()), "carriers with y=1" if not cont else "", int((K[i] * (y > 0)).sum()), "gpu flags", hex(int(r["flags"][i])))
    for f in fields:
        print("   %-9s gpu %.12g | oracle variants " % (f, r[f][i]) + " ".join("%.12g" % a[f][i] for a in alts))
    if kind == "fixed":
        print("   notes oracle", [hex(int(a["notes"][i])) for a in alts])
    else:
        print("   status oracle", [int(a["status"][i]) for a in alts], "fitll", [float(a["fitll"][i]) for a in alts])
