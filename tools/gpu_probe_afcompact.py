"""Development probe: LMM batch time at N = 5000 when a fraction of the k-mers is AF-filtered (real k-mer tables are U-shaped)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyseer_amd.engine import Engine, row_bytes_for
from bench import synth_lmm_inputs, synth_bits
N, V = 5000, 1 << 19
dev = torch.device("cuda", 0)
U, S, h2, C, y, _ = synth_lmm_inputs(N, 1003, dev)
bits = synth_bits(V, N, row_bytes_for(N), 11, dev)
for frac in (0.0, 0.02, 0.1, 0.3, 0.6):
    b = bits.clone()
    nr = int(V * frac)
    if nr:
        rare = torch.randperm(V, device=dev)[:nr]
        b[rare] = 0; b[rare, 0] = 1                                  # one carrier: AF = 1/N < 1 %
    for on in ("0", "1"):
        os.environ["SEERHIP_ROUTE"] = "afcompact=" + on
        e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
        e.lmm_setup(U, S, y, C, h2)
        for _ in range(4): e.lmm_batch_dev(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): e.lmm_batch_dev(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("extra filtered %.0f%%  compaction %s: %.2f ms per %d variants (%.1f M/s)" % (100 * frac, on, dt * 1e3, V, V / dt / 1e6))
        e.close()

# ---- fixed effects (logistic, q = 10): filtered lanes idle inside their wavefronts unless the rows are compacted
from pyseer_amd.model import fit_null
rng = np.random.default_rng(1002)
q = 10
W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
yb = (rng.random(N) < 1.0 / (1.0 + np.exp(0.3 - 1.5 * W[:, 0] + W[:, 1]))).astype(np.float64)
e0 = np.zeros((0, 0)); nl = fit_null(yb, W, e0, False).llf; nf = fit_null(yb, W, e0, False, firth=True)
Vg = 3 << 17
for frac in (0.0, 0.1, 0.3, 0.6):
    b = bits[:Vg].clone()
    nr = int(Vg * frac)
    if nr:
        rare = torch.randperm(Vg, device=dev)[:nr]
        b[rare] = 0; b[rare, 0] = 1
    for on in ("0", "1"):
        os.environ["SEERHIP_ROUTE"] = "afcompact=" + on
        e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
        e.glm_setup(yb, W, False, nl, nf)
        for _ in range(4): e.glm_batch_dev(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): e.glm_batch_dev(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("logistic: extra filtered %.0f%%  compaction %s: %.2f ms per %d variants (%.1f M/s)" % (100 * frac, on, dt * 1e3, Vg, Vg / dt / 1e6))
        e.close()

# ---- --filter-pvalue: almost every row fails the prefilter and is never fitted
for pret in (1.0, 1e-2, 1e-4):
    for on in ("0", "1"):
        os.environ["SEERHIP_ROUTE"] = "afcompact=" + on
        e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
        e.glm_setup(yb, W, False, nl, nf, pret, 1.0)
        b = bits[:Vg]
        for _ in range(4): e.glm_batch_dev(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): out_, fl_ = e.glm_batch_dev(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        pf = float(((fl_ >> 16) & 1).float().mean().item())
        print("logistic --filter-pvalue %g (%.1f%% prefiltered)  compaction %s: %.2f ms per %d variants (%.1f M/s)" % (pret, 100 * pf, on, dt * 1e3, Vg, Vg / dt / 1e6))
        e.close()

# ---- --filter-pvalue with --lmm: pre-filtered variants never reach the quadratic form
for pret in (1.0, 1e-2):
    for on in ("0", "1"):
        os.environ["SEERHIP_ROUTE"] = "afcompact=" + on
        e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
        e.lmm_setup(U, S, y, C, h2, continuous=False, filter_pvalue=pret, lrt_pvalue=1.0)
        for _ in range(4): e.lmm_batch_dev(bits)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): e.lmm_batch_dev(bits)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("LMM --filter-pvalue %g  compaction %s: %.2f ms per %d variants (%.1f M/s)" % (pret, on, dt * 1e3, V, V / dt / 1e6))
        e.close()
