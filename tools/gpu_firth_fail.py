import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyseer_amd.engine import Engine, row_bytes_for
from pyseer_amd.model import fit_null
from pyseer_amd.packing import unpack_variants
from bench import synth_bits
N, q, V = 1000, 10, 1 << 15
rng = np.random.default_rng(1002)
W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
eta = -0.3 + 1.5 * W[:, 0] - W[:, 1]
y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
e0 = np.zeros((0, 0))
nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
e = Engine(N); e.set_af_filter(0.01, 0.99)
e.glm_setup(y, W, False, nl, nf, force_firth=True)
bits = synth_bits(V, N, row_bytes_for(N), 5, torch.device("cuda")).cpu().numpy()
r = e.glm_batch(bits)
fail = np.where((r["flags"] >> 6) & 1)[0]
print("fails", len(fail), fail[:10])
K = unpack_variants(bits[fail[:8]], N)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/firth_fail.npz", y=y, W=W, K=K, nl=nl, nf=nf)
from oracle import oracle as orc
o = orc.firth_batch(y, K.astype(float), W)
print("oracle status", o["status"], o["kbeta"])
print("counts", K.sum(1), [y[K[i] == 1].sum() for i in range(len(K))])
