"""Development probe: throughput of the wide-design (q > 14) fixed-effects kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyseer_amd.engine import Engine, row_bytes_for
from pyseer_amd.model import fit_null
from bench import synth_bits
for N, q, V in ((1000, 20, 1 << 15), (5000, 20, 1 << 14), (5000, 32, 1 << 13)):
    rng = np.random.default_rng(5)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.5 * W[:, 0] - W[:, 1]
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf)
    bits = synth_bits(V, N, row_bytes_for(N), 5, torch.device("cuda"))
    e.glm_batch_dev(bits); torch.cuda.synchronize()
    t0 = time.time(); e.glm_batch_dev(bits); torch.cuda.synchronize(); dt = time.time() - t0
    print("N=%d q=%d V=%d: %.1f ms -> %.3g variants/s" % (N, q, V, dt * 1e3, V / dt))
    e.close()
for N, q, V in ((1000, 20, 1 << 16), (5000, 20, 1 << 16), (5000, 32, 1 << 15)):           # continuous phenotype (OLS)
    rng = np.random.default_rng(6)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    y = -0.3 + 1.5 * W[:, 0] - W[:, 1] + rng.standard_normal(N)
    nl = fit_null(y, W, np.zeros((0, 0)), True).llf
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, True, nl, None)
    bits = synth_bits(V, N, row_bytes_for(N), 5, torch.device("cuda"))
    e.glm_batch_dev(bits); torch.cuda.synchronize()
    t0 = time.time(); e.glm_batch_dev(bits); torch.cuda.synchronize(); dt = time.time() - t0
    print("OLS N=%d q=%d V=%d: %.1f ms -> %.3g variants/s" % (N, q, V, dt * 1e3, V / dt))
    e.close()
