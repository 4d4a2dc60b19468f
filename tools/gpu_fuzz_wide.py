"""One-off fuzz: k_lmm_quadform_i8w (SEERHIP_ROUTE qf=4, the default) against k_lmm_quadform_i8 (SEERHIP_ROUTE qf=0) on random shapes -- every output double
must be identical (the two kernels are the same exact integer contraction; tests/test_lmm_gpu.py holds six fixed shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_lmm_gpu import _random_lmm
from pyseer_amd.engine import Engine, pack_variants

rng = np.random.default_rng(int(os.environ.get("SEED", 11)))
bad = 0
cases = int(os.environ.get("CASES", 40))
for c in range(cases):
    N = int(rng.choice([512, 513, 640, 641, 767, 768, 769, 1024, 1279, 1280, 1500, 2047, 2048, 2500, 3000, 4095, 4096, 5000, 5121])) if c % 2 else int(rng.integers(512, 3000))
    V = int(rng.choice([1, 31, 511, 512, 513, 1000, 4096, 4097, 6000]))
    D = int(rng.choice([1, 1, 3]))
    limbs = int(rng.choice([0, 4, 5, 6]))
    U, S, covar, y, Kv = _random_lmm(N, D, 1000 + c, V)
    if V > 8:
        Kv[: V // 8] = (rng.random((V // 8, N)) < 0.93).astype(np.uint8)
    bits = pack_variants(Kv)
    out = []
    for qf in ("4", "0"):
        os.environ["SEERHIP_ROUTE"] = "qf=" + qf
        e = Engine(N); e.lmm_setup(U, S, y, covar, 0.3, n_limbs=limbs); out.append(e.lmm_batch(bits)); e.close()
    a, b = out
    same = np.array_equal(a["flags"], b["flags"]) and all(np.array_equal(a[f].view(np.uint64), b[f].view(np.uint64)) for f in ("prep", "beta", "bse", "pvalue", "frac_h2"))
    rows_last = N - (2 * ((N + 255) // 256) - 1) * 128
    print("case %2d N %5d (last tile rows %4d) V %5d D %d limbs %d: %s" % (c, N, rows_last, V, D, limbs, "identical" if same else "DIFFERENT"), flush=True)
    bad += not same
print("fuzz: %d of %d cases differ" % (bad, cases))
sys.exit(1 if bad else 0)
