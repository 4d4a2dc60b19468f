#!/usr/bin/env python
"""bench.py -- k-mer tests/second of the LMM hot path at N = 5000 samples on MI355X (BASELINE.json config C3).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the LMM per-variant path (sh_lmm_batch_dev: repack -> linear terms -> int8-MFMA quadratic
form -> finalise) over one batch of --variants-per-step synthetic packed k-mer presence rows that already live in HBM.
The k-mer stream shards across ranks with no collective on the data path (weak scaling: every rank tests its own
--variants-per-step rows per step).  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_SAMPLES = 5000
INT8_DENSE_PEAK_TOPS = 5000.0      # gfx950 int8 MFMA dense (2x the ~2.5 PF bf16 dense peak, MI355X_MICROARCH.md)
FP64_FLOP_PER_TEST = 5.0e7         # 2*k*N + 6k fp64 flop of the reference formulation, k=4999
ALGO_BYTES_PER_TEST = 673          # ceil(N/8) in + 48 out


def synth_lmm_inputs(N, seed, device):
    """C3 of SURVEY.md §8(d): K = G G^T from lineage-structured binary markers, scaled N/trace; D = 1; binary phenotype
    with a heritable component; decomposition + h2 by the package's own initialise_lmm restatement."""
    import torch
    from pyseer_amd.lmm import initialise_lmm_arrays
    rng = np.random.default_rng(seed)
    nlin, nmark = 40, 2000
    lin = rng.integers(0, nlin, N)
    base = rng.uniform(0.05, 0.95, (nlin, nmark))
    G = (rng.random((N, nmark)) < base[lin]).astype(np.float64)
    Gt = torch.from_numpy(G).to(device)
    K = (Gt @ Gt.T).cpu().numpy()
    g = G[:, :60].dot(rng.standard_normal(60)); g = (g - g.mean()) / g.std()
    y = ((0.7 * g + 0.7 * rng.standard_normal(N)) > 0).astype(np.float64)
    U, S, h2, nll, C = initialise_lmm_arrays(K, y, None, use_gpu=True)
    return U, S, h2, C, y, lin


def synth_bits(V, N, row_bytes, seed, device, chunk=1 << 16):
    """Packed presence rows, variant-major, generated on the device: AF ~ U(0.02, 0.98) (96 %), rare (2 %), i.i.d.
    Bernoulli(AF) presence.  Returns a (V, row_bytes) uint8 tensor resident in HBM."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    out = torch.zeros((V, row_bytes), dtype=torch.uint8, device=device)
    w = (2 ** torch.arange(8, device=device, dtype=torch.int32)).to(torch.uint8)
    nb = (N + 7) // 8
    for s in range(0, V, chunk):
        n = min(chunk, V - s)
        af = torch.rand((n, 1), generator=g, device=device) * 0.96 + 0.02
        rare = torch.rand((n, 1), generator=g, device=device) < 0.02
        af = torch.where(rare, af * 0.02, af)
        x = torch.rand((n, nb * 8), generator=g, device=device) < af
        x[:, N:] = False
        out[s:s + n, :nb] = (x.view(n, nb, 8).to(torch.uint8) * w).sum(dim=2, dtype=torch.uint8)
    return out


def cpu_baseline(U, S, y, C, h2, N, target_s=12.0):
    """The CPU oracle (oracle/seer_oracle.c: a C port of the reference algorithm, OpenMP over variants) on a bounded
    sample of the same workload, timed on this host's cores."""
    from oracle import oracle as orc
    ncores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
    L = orc.LmmOracle(U, S, y, C)
    rng = np.random.default_rng(77)

    def run(v):
        af = rng.uniform(0.02, 0.98, v)
        Kv = (rng.random((v, N)) < af[:, None]).astype(np.float64)
        t0 = time.time(); L.block(h2, Kv); return time.time() - t0
    v0 = 8 * ncores
    t = run(v0)
    v1 = int(min(max(v0, v0 * target_s / max(t, 1e-3)), 200000))
    v1 = max(8, (v1 // 8) * 8)
    t1 = run(v1)
    return dict(value=v1 / t1, unit="variants/s", cores=ncores, kind="port",
                sample="%d synthetic k-mers x %d samples, LMM block test (oracle/seer_oracle.c orc_lmm_block, OpenMP)" % (v1, N))


def fixed_effects_extra(dev, local, N, q=10, V=1 << 18, reps=3):
    """Secondary numbers for the fixed-effects half of the metric (BASELINE configs C2/C4 shapes at N samples):
    logistic with 10 MDS-like covariates, and Firth forced on every variant.  Same packed-bit inputs, resident in HBM."""
    import torch
    from pyseer_amd.engine import Engine, row_bytes_for
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(1002)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.5 * W[:, 0] - W[:, 1]
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-eta))).astype(np.float64)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf
    nf = fit_null(y, W, e0, False, firth=True)
    out = {}
    for name, force, v in (("logistic", False, V), ("firth", True, V)):
        eng = Engine(N, device=local); eng.use_torch_stream(); eng.set_af_filter(0.01, 0.99)
        eng.glm_setup(y, W, False, nl, nf, 1.0, 1.0, force_firth=force)
        bits = synth_bits(v, N, row_bytes_for(N), 4242, dev)
        o = torch.empty((5 + q, v), dtype=torch.float64, device=dev); f = torch.empty((v,), dtype=torch.int32, device=dev)
        eng.glm_batch_dev(bits, o, f); torch.cuda.synchronize()
        eng.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.glm_batch_dev(bits, o, f)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        kms, kl = eng.get_timing()
        # SURVEY.md §8(d) work model: logistic ~1.1e6 fp64 flop/test at N=1000 (x N/1000), Firth ~9e6 at N=5000
        flop = (1.1e6 * N / 1000.0) if not force else (9.0e6 * N / 5000.0)
        out[name] = {"variants_per_s": v / dt, "n_samples": N, "q": q, "variants": v,
                     "dominant_kernel": "k_firth_eval + k_firth_step (rounds)" if force else "k_glm_fast + k_glm_slow + k_glm_final", "kernel_ms": kms / max(kl, 1),
                     "fp64_vector_tflops_model": flop * v / (kms / max(kl, 1) * 1e-3) / 1e12, "fp64_vector_peak_tflops": 78.6}
        eng.close()
    return out


def measured_traffic(Vs):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (they cannot run inside
    this process: counters need their own rocprofv3 runs), scaled by variants per launch.  None if no measurement is committed."""
    path = os.path.join(ROOT, "profiles", "r01", "traffic_lmm.json")
    if not os.path.exists(path):
        return None, None
    t = json.load(open(path))
    per_variant = (t["FETCH_SIZE_KB"] * t["fetch_correction"] + t["WRITE_SIZE_KB"]) * 1024.0 / t["variants_per_dispatch"]
    return per_variant * Vs, t["source"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--variants-per-step", type=int, default=1 << 20)
    ap.add_argument("--limbs", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary fixed-effects measurements")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pyseer_amd.engine import Engine, row_bytes_for
    N = N_SAMPLES
    U, S, h2, C, y, _ = synth_lmm_inputs(N, 1003, dev)          # identical per-run constants on every rank (replicated)
    eng = Engine(N, device=local)
    eng.use_torch_stream()
    eng.lmm_setup(U, S, y, C, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0, n_limbs=args.limbs)
    info = eng.lmm_info()
    rb = row_bytes_for(N)
    Vs = args.variants_per_step
    nbuf = args.steps + args.warmup
    # every step has its own rows (seed + rank shard), all resident in HBM before the timed region
    bits = [synth_bits(Vs, N, rb, 1003 + 1000 * rank + i, dev) for i in range(nbuf)]
    out = torch.empty((5, Vs), dtype=torch.float64, device=dev)
    fl = torch.empty((Vs,), dtype=torch.int32, device=dev)

    for i in range(args.warmup):
        eng.lmm_batch_dev(bits[i], out, fl)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    eng.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eng.lmm_batch_dev(bits[args.warmup + i], out, fl)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    kms, klaunch = eng.get_timing()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # sanity: the timed work produced finite statistics
    frac_finite = float(torch.isfinite(out[2]).double().mean().item())

    if rank == 0:
        total = float(Vs) * args.steps * world
        value = total / dt
        kern_s = kms / max(klaunch, 1) * 1e-3
        int8_ops = 2.0 * info["int8_macs_per_variant"] * Vs
        achieved = int8_ops / kern_s / 1e12
        traffic, traffic_src = measured_traffic(Vs)
        res = {
            "metric": "k-mer tests/sec at N=5000 samples (LMM), whole job",
            "value": value, "unit": "variants/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8 limbs x%d (exact int32 MFMA accumulation, fp64 recombination and statistics)" % info["n_limbs"],
            "data": "synthetic",
            "config": {"workload": "C3: LMM (FaST-LMM per-variant test), %d synthetic k-mers x %d samples per step per GPU, "
                                   "D=1, k=%d, h2=%.4f, inputs resident in HBM" % (Vs, N, U.shape[1], h2),
                       "variants_per_step_per_gpu": Vs, "n_samples": N, "sharding": "k-mer stream sharded by rank, no collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TFLOP/s",
                         "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "kernel": "k_lmm_quadform_i8", "kernel_ms": kern_s * 1e3, "launches": klaunch,
                         "ops": "int8 multiply-adds x2 actually issued: L*NR*(NR+1)*128*64 per variant, NR = ceil(N/128)",
                         "fp64_equiv_tflops": FP64_FLOP_PER_TEST * Vs / kern_s / 1e12,
                         "hbm_algorithmic_GBps": ALGO_BYTES_PER_TEST * Vs / kern_s / 1e9},
            "finite_fraction": frac_finite,
        }
        if world == 1 and not args.no_extra:
            res["extra"] = {"fixed_effects_N5000": fixed_effects_extra(dev, local, N_SAMPLES),
                            "fixed_effects_N1000": fixed_effects_extra(dev, local, 1000, V=1 << 20)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(U, S, y, C, h2, N)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
