"""Development probe: throughput of fit_lineage_effect on the GPU for narrow (registers) and wide (run-time width) lineage designs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyseer_amd.engine import Engine, pack_variants
rng = np.random.default_rng(3)
for N, nl, V in ((5000, 9, 65536), (5000, 15, 32768), (5000, 30, 4096), (1000, 30, 8192)):
    cl = rng.integers(0, nl + 1, N)
    lin = np.zeros((N, nl)); lin[np.arange(N)[cl > 0], cl[cl > 0] - 1] = 1.0
    base = rng.uniform(0.3, 0.7, (256, nl + 1))
    K = (rng.random((256, N)) < base[:, cl]).astype(np.uint8)
    bits = np.tile(pack_variants(K), (V // 256, 1))
    e = Engine(N); e.lineage_setup(lin, None)
    e.lineage_batch(bits[:256])
    t0 = time.time(); e.lineage_batch(bits); dt = time.time() - t0
    print("lineage N=%d clusters=%d V=%d: %.1f ms -> %.3g variants/s" % (N, nl, V, dt * 1e3, V / dt))
    e.close()
