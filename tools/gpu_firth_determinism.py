"""Is a forced-Firth batch the same bits every time?  For each route of tests/test_glm_gpu.py::test_one_pass_firth_equals_the_two_pass_rounds and each
of its shapes: REP synchronous calls on one context (and REP more with another stream of the device kept busy), every output compared bit for bit
with the first.  A kernel with a timing-dependent read (a ring stage read before it landed, a missing wait) shows up as a run that differs."""
import json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyseer_amd.engine import Engine, pack_variants
from pyseer_amd.model import fit_null

REP = int(os.environ.get("REP", 40))
SHAPES = [(4100, 1, 1024), (4099, 3, 777), (5000, 10, 8192), (6007, 7, 1500), (4096, 10, 640), (8200, 5, 512), (4100, 2, 1024)]
ROUTES = [("two", "firth_fast=0"), ("one", None), ("one_w0", "firth_w=0"), ("one_f64", "firth_first32=0"), ("one_r5", "firth_first32=1")]
stop = False


def busy():
    a = torch.randn(4096, 4096, device="cuda"); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(10):
                a = (a @ a).clamp(-1, 1)
            s.synchronize()


res = []
for N, q, V in SHAPES:
    rng = np.random.default_rng(97 + N + q)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.2 * W[:, 0] - (0.7 * W[:, 1] if q > 1 else 0.0)
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    af = np.concatenate([rng.uniform(0.02, 0.98, V - V // 4), rng.uniform(0.0101, 0.03, V // 8), rng.uniform(0.97, 0.9899, V // 4 - V // 8)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    K[: V // 10] = (rng.random((V // 10, N)) < (0.15 + 0.5 * y)[None, :]).astype(np.uint8)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    bits = pack_variants(K)
    for mode, route in ROUTES:
        if route is None:
            os.environ.pop("SEERHIP_ROUTE", None)
        else:
            os.environ["SEERHIP_ROUTE"] = route
        e = Engine(N); e.set_af_filter(0.01, 0.99); e.glm_setup(y, W, False, nl, nf, force_firth=True)
        first = None; differ = 0; worst = 0.0; flagdiff = 0
        for phase in (0, 1):
            if phase == 1:
                stop = False; th = threading.Thread(target=busy); th.start()
            for r in range(REP):
                o = e.glm_batch(bits)
                cur = {k: np.array(v, copy=True) for k, v in o.items() if isinstance(v, np.ndarray)}
                if first is None:
                    first = cur; continue
                same = all(np.array_equal(first[k].view(np.uint8), cur[k].view(np.uint8)) for k in first)
                if not same:
                    differ += 1
                    okm = np.isfinite(first["kbeta"]) & np.isfinite(cur["kbeta"])
                    worst = max(worst, float(np.max(np.abs(first["kbeta"][okm] - cur["kbeta"][okm]) / (1e-6 * np.abs(first["kbeta"][okm]) + 2e-8))) if okm.any() else 0.0)
                    flagdiff += int((first["flags"] != cur["flags"]).sum())
            if phase == 1:
                stop = True; th.join()
        e.close()
        r_ = {"N": N, "q": q, "V": V, "mode": mode, "runs": 2 * REP, "runs_that_differ": differ, "worst_kbeta_tol_units": worst, "flag_differences": flagdiff}
        res.append(r_); print(json.dumps(r_), flush=True)
os.environ.pop("SEERHIP_ROUTE", None)
print("ANY_DIFFERENCE", any(r["runs_that_differ"] for r in res))
