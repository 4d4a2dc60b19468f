"""Similarity (kinship) accumulation rate: N samples x V variants resident in HBM (env N, V)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyseer_amd.engine import Engine

N = int(os.environ.get("N", 5000)); V = int(os.environ.get("V", 1 << 20))
rb = (N + 63) // 64 * 8
g = torch.Generator(device="cuda"); g.manual_seed(1)
bits = torch.randint(0, 256, (V, rb), dtype=torch.uint8, device="cuda", generator=g)
if N % 8 or rb * 8 != N:
    mask = np.zeros(rb, dtype=np.uint8); full = N // 8
    mask[:full] = 255
    if N % 8: mask[full] = (1 << (N % 8)) - 1
    bits &= torch.from_numpy(mask).cuda()
e = Engine(N)
e.sim_begin(); e.sim_accumulate_dev(bits); e.synchronize()
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    e.sim_accumulate_dev(bits)
e.synchronize()
dt = (time.perf_counter() - t0) / reps
pairs = N * (N + 1) / 2 * V
print("N=%d V=%d: %.1f ms per batch  %.3g variants/s  %.3g sample-pair-variants/s (%.2f T 64-bit AND+popcount/s)"
      % (N, V, dt * 1e3, V / dt, pairs / dt, pairs / 64 / dt / 1e12))
K = e.sim_finish()
# spot check against a dense product on a slice
sub = bits[:, :8].cpu().numpy()
d = np.unpackbits(sub, axis=1, bitorder="little")[:, :64].astype(np.float64)
ref = d.T @ d * (reps + 1)
print("spot check 64x64 corner exact:", bool(np.array_equal(K[:64, :64], ref)))
