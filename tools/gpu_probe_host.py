"""Development probe: host-pointer entry point (PCIe-inclusive) and dedup overhead/benefit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyseer_amd.engine import Engine, row_bytes_for
from bench import synth_bits
N, V = 5000, 1 << 20
g = torch.Generator(device="cuda"); g.manual_seed(5)
U = (torch.randn((N, N - 1), generator=g, device="cuda", dtype=torch.float64) / np.sqrt(N)).cpu().numpy()
rng = np.random.default_rng(9)
S = np.sort(rng.gamma(0.5, 2.0, N - 1))[::-1].copy()
y = (rng.random(N) < 0.4).astype(float); covar = np.ones((N, 1))
e = Engine(N); e.use_torch_stream(); e.lmm_setup(U, S, y, covar, 0.3)
bits_d = synth_bits(V, N, row_bytes_for(N), 11, torch.device("cuda"))
bits_h = bits_d.cpu().numpy()
for name, fn in (("device-resident", lambda: (e.lmm_batch_dev(bits_d), torch.cuda.synchronize())), ("host-pointer (PCIe-inclusive)", lambda: e.lmm_batch(bits_h))):
    fn(); t0 = time.time(); fn(); dt = time.time() - t0
    print("%-32s %.1f ms  %.3g variants/s" % (name, dt * 1e3, V / dt))
# dedup: all-unique input (pure overhead) and 8x duplicated input (benefit)
e.set_dedup(True)
e.lmm_batch_dev(bits_d); torch.cuda.synchronize(); t0 = time.time(); e.lmm_batch_dev(bits_d); torch.cuda.synchronize(); dt = time.time() - t0
print("dedup on, all unique (%d): %.1f ms  %.3g variants/s" % (e.dedup_info(), dt * 1e3, V / dt))
dup = bits_d[torch.randint(0, V // 8, (V,), device="cuda", generator=g)].contiguous()
e.lmm_batch_dev(dup); torch.cuda.synchronize(); t0 = time.time(); e.lmm_batch_dev(dup); torch.cuda.synchronize(); dt = time.time() - t0
print("dedup on, %d unique of %d: %.1f ms  %.3g variants/s" % (e.dedup_info(), V, dt * 1e3, V / dt))
