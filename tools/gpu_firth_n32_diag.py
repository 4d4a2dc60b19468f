"""Development aid (round 6): the one-pass Firth path with one / two single-precision first passes (SEERHIP_ROUTE firth_first32=1 / 2) against the
exact two-pass rounds on the C4 workload: the rows where kbeta deviates most, with what distinguishes them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for
N, q, V = 5000, 10, int(os.environ.get("V", 262144))
y, W, nl, nf = bench.synth_glm_inputs(N, q)
dev = torch.device("cuda", 0)
bits = bench.synth_bits(V, N, row_bytes_for(N), 4242, dev)
outs = {}
for mode, route in (("two", "firth_fast=0"), ("n1", "firth_first32=1"), ("n2", "firth_first32=2")):
    os.environ["SEERHIP_ROUTE"] = route
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf, force_firth=True)
    o, f = e.glm_batch_dev(bits); torch.cuda.synchronize()
    outs[mode] = (o.cpu().numpy(), f.cpu().numpy())
    e.close()
K = bench.unpack_rows(bits.cpu().numpy(), N)
af = K.mean(axis=1)
a = outs["two"][0]
ok = np.isfinite(a[2])
for mode in ("n1", "n2"):
    b = outs[mode][0]
    d = np.where(ok, np.abs(b[2] - a[2]), 0)
    tolu = d / (1e-6 * np.abs(a[2]) + 2e-8)
    print("== %s: kbeta max abs %.3g, max in tolerance units %.3f, rows over 0.25 of the tolerance %d; bse max rel %.3g" % (
        mode, d.max(), np.nanmax(np.where(ok, tolu, 0)), int((np.where(ok, tolu, 0) > 0.25).sum()),
        np.nanmax(np.where(ok, np.abs(b[3] - a[3]) / a[3], 0))))
    for i in np.argsort(-np.where(ok, tolu, 0))[:8]:
        print("   row %6d af %.4f kbeta %.6g bse %.4g  d %.3g (%.2f tol units)  carriers %d  y1 among carriers %d" % (
            i, af[i], a[2][i], a[3][i], d[i], tolu[i], int(K[i].sum()), int((K[i] * y).sum())))
