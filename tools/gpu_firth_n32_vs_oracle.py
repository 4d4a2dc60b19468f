"""Round 6: the one-pass Firth path with one / two single-precision first passes (SEERHIP_ROUTE firth_first32=1 / 2) against the ORACLE
(orc_firth_batch: fit_firth restated, pinned to the reference) on V rows of the bench's C4 workload: deviations of kbeta / bse / intercept / p
under the suite's rule for Firth rows (1e-6 relative or 2e-8 absolute) and purely relative; flags."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import oracle as orc
from pyseer_amd.engine import Engine, row_bytes_for
N, q, V = 5000, 10, int(os.environ.get("V", 32768))
orc.set_threads(min(64, len(os.sched_getaffinity(0))))
y, W, nl, nf = bench.synth_glm_inputs(N, q)
dev = torch.device("cuda", 0)
bits = bench.synth_bits(V, N, row_bytes_for(N), 4242, dev)
K = bench.unpack_rows(bits.cpu().numpy(), N)
af = K.mean(axis=1); keep = (af >= 0.01) & (af <= 0.99)
t0 = time.time(); w = orc.firth_batch(y, K, W); t_or = time.time() - t0
lr = -2.0 * (nf - w["fitll"]); wp = np.array([orc.chi2_sf1(x) if x > 0 else 1.0 for x in lr])
res = {"rows": V, "oracle_s": round(t_or, 1)}
for n in ("1", "2"):
    os.environ["SEERHIP_ROUTE"] = "firth_first32=" + n
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf, force_firth=True)
    o, f = e.glm_batch_dev(bits); torch.cuda.synchronize()
    o = o.cpu().numpy(); f = f.cpu().numpy().astype(np.uint32); e.close()
    ok = keep & (w["status"] == 0) & np.isfinite(o[2])
    r = {"compared": int(ok.sum()), "firth_fail_here": int(((f >> 6) & 1)[keep].sum()), "oracle_fail": int((w["status"][keep] != 0).sum())}
    for name, row, want in (("kbeta", 2, w["kbeta"]), ("bse", 3, w["bse"]), ("intercept", 4, w["intercept"]), ("pvalue", 1, wp)):
        d = np.abs(o[row][ok] - want[ok])
        r[name] = {"max_abs": float(d.max()), "max_rel": float(np.max(d / np.maximum(np.abs(want[ok]), 1e-300))),
                   "rows_over_1e-6_rel": int((d > 1e-6 * np.abs(want[ok])).sum()),
                   "max_in_units_of_1e-6rel_plus_2e-8": float(np.max(d / (1e-6 * np.abs(want[ok]) + 2e-8)))}
    res["firth_first32=" + n] = r
print(json.dumps(res, indent=1))
