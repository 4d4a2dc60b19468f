"""First-contact GPU probe: time the LMM pipeline at N=5000 (not the bench; a development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyseer_amd.engine import Engine, row_bytes_for

N = int(os.environ.get("N", 5000)); V = int(os.environ.get("V", 65536)); L = int(os.environ.get("L", 5))
g = torch.Generator(device="cuda"); g.manual_seed(5)
U = (torch.randn((N, N - 1), generator=g, device="cuda", dtype=torch.float64) / np.sqrt(N)).cpu().numpy()
rng = np.random.default_rng(9)
S = np.sort(rng.gamma(0.5, 2.0, N - 1))[::-1].copy()
D = int(os.environ.get("D", 1)); CONT = int(os.environ.get("CONT", 0))
y = rng.standard_normal(N) if CONT else (rng.random(N) < 0.4).astype(float)
covar = np.ones((N, 1)) if D == 1 else np.c_[rng.standard_normal((N, D - 1)), np.ones((N, 1))]
e = Engine(N)
t0 = time.time(); e.lmm_setup(U, S, y, covar, 0.3, continuous=bool(CONT), n_limbs=L); print("setup s", time.time() - t0, e.lmm_info())
rb = row_bytes_for(N)
bits = torch.randint(0, 256, (V, rb), dtype=torch.uint8, device="cuda", generator=g)
e.use_torch_stream(); e.set_timing(True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out, fl = e.lmm_batch_dev(bits)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("iter", it, "V", V, "s", dt, "variants/s", V / dt)
ms, n = e.get_timing()
info = e.lmm_info()
print("quadform ms/launch", ms / n, "int8 TOPS", 2 * info["int8_macs_per_variant"] * V / (ms / n * 1e-3) / 1e12)
print(out[:, :4].cpu().numpy())
