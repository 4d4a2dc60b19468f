"""Development aid (round 6, review item 1b): where does the logistic path's bse deviate from the oracle's?  A large randomised sweep in chunks
(rows as tools/gpu_glm_sweep.py draws them: a binary and an un-centred covariate, 15 % real / near-separating effects), per-row relative
deviations kept; prints quantiles per field, the worst rows with what distinguishes them (allele frequency, |kbeta|, kbeta / bse, the
minor-allele count), and the deviation binned by allele frequency and by effect size.  N, Q, V, SEED, CHUNK from the environment."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from pyseer_amd.engine import Engine, pack_variants
from pyseer_amd.model import fit_null

N = int(os.environ.get("N", 5000)); q = int(os.environ.get("Q", 10)); V = int(os.environ.get("V", 100000)); seed = int(os.environ.get("SEED", 11))
CHUNK = int(os.environ.get("CHUNK", 20000))
orc.set_threads(int(os.environ.get("THREADS", os.cpu_count() and min(16, os.cpu_count()) or 4)))
rng = np.random.default_rng(seed)
W = rng.standard_normal((N, q)); W[:, 0] = rng.random(N) < 0.3
if q > 1: W[:, 1] = float(os.environ.get("OFFSET", 2000)) + 10 * W[:, 1]
eta = -0.5 + 0.9 * W[:, 0] + 0.5 * W[:, min(2, q - 1)]
y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
e0 = np.zeros((0, 0))
nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
e = Engine(N); e.glm_setup(y, W, False, nl, nf)
cols = {k: [] for k in ("af", "kbeta", "bse", "pvalue", "d_bse", "d_kbeta", "d_p", "d_int", "firth", "mism", "got_flags", "want_notes", "got_p", "got_bse", "got_kbeta")}
t_or = 0.0
done = 0
while done < V:
    v = min(CHUNK, V - done)
    af = np.concatenate([rng.uniform(0.02, 0.98, v // 2), rng.beta(0.3, 0.3, v - v // 2)])
    K = (rng.random((v, N)) < af[:, None])
    eff = rng.random(v) < 0.15
    K[eff] = rng.random((int(eff.sum()), N)) < (0.05 + 0.8 * (y > np.median(y)))[None, :] * rng.uniform(0.1, 1.0, int(eff.sum()))[:, None]
    K = K[(K.mean(axis=1) >= 0.01) & (K.mean(axis=1) <= 0.99)].astype(np.uint8)
    t0 = time.time(); want = orc.fixed_effects_batch(y, K.astype(float), W, False, 1.0, 1.0, nl, nf); t_or += time.time() - t0
    r = e.glm_batch(pack_variants(K))
    firth = (want["notes"] & 0x7C) != 0
    rel = lambda f: np.abs(r[f] - want[f]) / np.maximum(np.abs(want[f]), 1e-300)
    cols["af"].append(K.mean(axis=1)); cols["kbeta"].append(want["kbeta"]); cols["bse"].append(want["bse"]); cols["pvalue"].append(want["pvalue"])
    cols["d_bse"].append(rel("bse")); cols["d_kbeta"].append(rel("kbeta")); cols["d_p"].append(rel("pvalue")); cols["d_int"].append(rel("intercept"))
    cols["firth"].append(firth); cols["mism"].append((r["flags"] & 0x1FF) != want["notes"])
    cols["got_flags"].append(r["flags"] & 0x1FF); cols["want_notes"].append(want["notes"]); cols["got_p"].append(r["pvalue"]); cols["got_bse"].append(r["bse"]); cols["got_kbeta"].append(r["kbeta"])
    done += v
e.close()
c = {k: np.concatenate(v) for k, v in cols.items()}
ok = ~c["firth"] & np.isfinite(c["bse"]) & np.isfinite(c["d_bse"])
out = {"N": N, "q": q, "rows": int(len(ok)), "newton_rows": int(ok.sum()), "firth_rows": int(c["firth"].sum()), "note_mismatches": int(c["mism"].sum()),
       "oracle_s": round(t_or, 1)}
for f in ("d_bse", "d_kbeta", "d_p", "d_int"):
    x = c[f][ok & np.isfinite(c[f])]
    out[f] = {"max": float(x.max()), "q999": float(np.quantile(x, 0.999)), "q99": float(np.quantile(x, 0.99)), "median": float(np.median(x))}
print(json.dumps(out))
d = np.where(ok, c["d_bse"], 0)
maf = np.minimum(c["af"], 1 - c["af"])
print("bse deviation by allele frequency (max / q99):")
for lo, hi in ((0, .02), (.02, .05), (.05, .2), (.2, .5), (.5, .8), (.8, .95), (.95, .98), (.98, 1)):
    m = ok & (c["af"] >= lo) & (c["af"] < hi)
    if m.any(): print("  af [%.2f, %.2f): n %7d  max %.2e  q99 %.2e  median %.2e" % (lo, hi, m.sum(), d[m].max(), np.quantile(d[m], .99), np.median(d[m])))
print("bse deviation by |kbeta| (max / q99):")
ab = np.abs(c["kbeta"])
for lo, hi in ((0, .05), (.05, .2), (.2, .5), (.5, 1), (1, 2), (2, 4), (4, 1e9)):
    m = ok & (ab >= lo) & (ab < hi)
    if m.any(): print("  |kbeta| [%.2f, %.2f): n %7d  max %.2e  q99 %.2e  median %.2e" % (lo, hi, m.sum(), d[m].max(), np.quantile(d[m], .99), np.median(d[m])))
for i in np.argsort(-d)[:15]:
    print("  worst bse row %d: af %.4f kbeta %.5g bse %.5g z %.3g p %.3g  d_bse %.2e d_kbeta %.2e d_p %.2e" % (
        i, c["af"][i], c["kbeta"][i], c["bse"][i], c["kbeta"][i] / c["bse"][i], c["pvalue"][i], c["d_bse"][i], c["d_kbeta"][i], c["d_p"][i]))
dp = np.where(ok & np.isfinite(c["d_p"]), c["d_p"], 0)
for i in np.argsort(-dp)[:6]:
    print("  worst p row %d: af %.4f kbeta %.6g/%.6g bse %.5g p %.12g/%.12g  d_p %.2e" % (i, c["af"][i], c["got_kbeta"][i], c["kbeta"][i], c["bse"][i], c["got_p"][i], c["pvalue"][i], c["d_p"][i]))
for i in np.flatnonzero(c["mism"])[:12]:
    print("  note mismatch row %d: got %#x want %#x  af %.4f  kbeta %.6g/%.6g bse %.9g/%.9g p %.4g/%.4g" % (i, c["got_flags"][i], c["want_notes"][i], c["af"][i],
          c["got_kbeta"][i], c["kbeta"][i], c["got_bse"][i], c["bse"][i], c["got_p"][i], c["pvalue"][i]))
