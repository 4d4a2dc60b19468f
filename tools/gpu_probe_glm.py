"""Development probe: throughput of the fixed-effects kernels (C2 / C4 shapes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyseer_amd.engine import Engine, row_bytes_for
from pyseer_amd.model import fit_null
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_bits

def run(N, q, V, force_firth, cont=False):
    rng = np.random.default_rng(1002)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.5 * W[:, 0] - W[:, 1]
    if os.environ.get("YEARCOL"):                       # an un-centred covariate (e.g. year of isolation): 2000 +- 10
        W = W.copy(); W[:, 2] = 2000.0 + 10.0 * W[:, 2]
    y = eta + rng.standard_normal(N) if cont else (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, cont).llf
    nf = np.nan if cont else fit_null(y, W, e0, False, firth=True)
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(*((0.0005, 0.9995) if os.environ.get("MIXED") else (0.01, 0.99)))
    e.glm_setup(y, W, cont, nl, nf, force_firth=force_firth)
    bits = synth_bits(V, N, row_bytes_for(N), 5, torch.device("cuda"))
    out, fl = e.glm_batch_dev(bits); torch.cuda.synchronize()
    e.set_timing(True)
    t0 = time.time(); out, fl = e.glm_batch_dev(bits); torch.cuda.synchronize(); dt = time.time() - t0
    ms, n = e.get_timing()
    f = fl.cpu().numpy().astype(np.uint32)
    print(e.glm_info(), "firth-fail", np.mean((f >> 6) & 1))
    print("N=%d q=%d V=%d firth=%d cont=%d: %.1f ms total, dominant kernel %.1f ms, %.3g variants/s ; firth-routed %.2f%%, prefiltered %.2f%%"
          % (N, q, V, force_firth, cont, dt * 1e3, ms, V / dt, 100 * np.mean((f & 0x7C) != 0), 100 * np.mean((f >> 16) & 1)))
    e.close()

if os.environ.get("LOGIT_V"):
    run(int(os.environ.get("N", 5000)), 10, int(os.environ["LOGIT_V"]), False)
elif os.environ.get("FIRTH_V"):
    run(int(os.environ.get("N", 5000)), 10, int(os.environ["FIRTH_V"]), True)
else:
    run(1000, 10, 1 << 18, False)
    run(5000, 10, 1 << 17, False)
    run(5000, 10, 1 << 15, True)
    run(1000, 10, 1 << 16, True)
    run(5000, 10, 1 << 18, False, cont=True)
