#!/usr/bin/env python
"""bench.py -- k-mer tests/second of the per-variant association hot path on MI355X (BASELINE.json configs C2 / C3 / C4).

    python bench.py --gpus 1 --steps K --warmup W [--config C3|C2|C2N5000|C4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic packed k-mer presence rows that already live in HBM:
  C3 (default)  LMM, N = 5000, D = 1: sh_lmm_batch_dev = repack -> linear terms -> int8-MFMA quadratic form -> finalise
  C2            fixed-effects logistic regression, N = 1000, 10 covariates: sh_glm_batch_dev (Newton kernels, Firth for routed rows)
  C2N5000       the same at N = 5000 (the "fixed-effects at N = 5000" half of BASELINE's metric)
  C4            N = 5000, 10 covariates, every variant through Firth (force_firth)
The k-mer stream shards across ranks with no collective on the data path (weak scaling: every rank tests its own
--variants-per-step rows per step).  One JSON line is printed by rank 0; it carries `roofline`, `cpu_baseline` (N = 1) and
`parity_checked` = number of variants of the TIMED output that were re-checked against the CPU oracle after the timed region.
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
FP64_VECTOR_PEAK_TFLOPS = 78.6     # gfx950 fp64 vector peak
# the reference itself on one thread (measured with statsmodels 0.12.2 / numpy in the build container), variants/s at N = 5000
REFERENCE_PER_CORE = {"logistic": 112.0, "firth": 1.0, "lmm": 303.0}
GLM_KERNELS = {False: "k_glm_fast<Q,true> (prefilter, routing) + k_glm_bitdot + k_glm_solve32 + k_glm_pass32(_split) + k_glm_score(_split) / k_glm_chord + k_glm_ll + "
                      "k_glm_dpass_pk + k_glm_finish (+ k_glm_slow_blk and the Firth kernels for routed rows)",
               True: "k_glm_fast<Q,true> + k_glm_bitdot + k_firth_init2 + passes of k_firth_fast (one sample pass per Firth iteration: eta / exp / log-likelihood / "
                     "exact score in fp64 on the vector ALU, the information matrix and the penalty's third-moment tensor as f16 hi/lo MFMAs with fp32 accumulation; "
                     "the first pass, at the start vector, in single precision with one f16 product per tile at two wavefronts per SIMD: k_firth_fast32w + "
                     "k_firth_fastw_fin; fits finished in the kernel) + k_firth_eval2 / "
                     "k_firth_step2 (exact two-pass rounds) for the fits that leave the fast passes"}
FP64_FLOP_PER_TEST = 5.0e7         # 2*k*N + 6k fp64 flop of the reference formulation, k=4999
ALGO_BYTES_PER_TEST = 673          # ceil(N/8) in + 48 out
PROFILE_DIRS = ("r06", "r05", "r04", "r03", "r02", "r01")      # committed rocprofv3 summaries, newest first


# ---------------------------------------------------------------------------------------------------------------
# clock and board power over the timed region (round-3 review, item 8): boxes of the pool differ by 10 % on the same binary, and the
# kernel runs into the chip's clock management, so the line says at what clock its fraction was measured
# ---------------------------------------------------------------------------------------------------------------
class ClockSampler(object):
    """Samples the amdgpu hwmon files of one device from a side thread every `period` seconds: freq1_input (sclk, Hz) and power1_average /
    power1_input (microwatts).  Reading sysfs does not touch the GPU's queues.  Everything is best effort: a box without the files gives
    nulls, never an error."""

    def __init__(self, torch_index=0, period=0.004):
        import glob
        self.period = period
        self.f_clk = self.f_pow = None
        self.clk, self.pow = [], []
        self._stop = False
        self._thread = None
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            if hw:
                cards.append((os.path.realpath(dev), hw[0]))
        if not cards:
            return
        pick = cards[min(torch_index, len(cards) - 1)]
        try:                                               # match torch's device by PCI address where torch exposes it
            import torch
            pr = torch.cuda.get_device_properties(torch_index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            for c in cards:
                if want in c[0]:
                    pick = c
        except Exception:
            pass
        self.source = pick[1]
        f = os.path.join(pick[1], "freq1_input")
        self.f_clk = f if os.path.exists(f) else None
        for nm in ("power1_average", "power1_input"):
            f = os.path.join(pick[1], nm)
            if os.path.exists(f):
                self.f_pow = f
                break

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop:
            if self.f_clk:
                v = self._read(self.f_clk)
                if v is not None:
                    self.clk.append(v)
            if self.f_pow:
                v = self._read(self.f_pow)
                if v is not None:
                    self.pow.append(v)
            time.sleep(self.period)

    def start(self):
        import threading
        self.clk, self.pow, self._stop = [], [], False
        if self.f_clk or self.f_pow:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        out = {"sclk_mhz_mean": None, "sclk_mhz_min": None, "power_w_mean": None, "power_w_max": None, "clock_samples": len(self.clk),
               "clock_source": ("sysfs hwmon freq1_input / %s, sampled every %.0f ms over the timed region" %
                                (os.path.basename(self.f_pow) if self.f_pow else "-", self.period * 1e3)) if (self.f_clk or self.f_pow) else None}
        if self.clk:
            out["sclk_mhz_mean"] = float(np.mean(self.clk)) / 1e6; out["sclk_mhz_min"] = float(np.min(self.clk)) / 1e6
        if self.pow:
            out["power_w_mean"] = float(np.mean(self.pow)) / 1e6; out["power_w_max"] = float(np.max(self.pow)) / 1e6
        return out


NOMINAL_SCLK_MHZ = 2400.0          # the clock the guide's peaks are quoted at (MI355X_MICROARCH.md)


def with_clock(roof, clocks):
    """roofline dict + the sampled clock / power, and the same fraction against the peak AT THE CLOCK THE CHIP GRANTED (frac itself stays
    against the nominal-clock peak): two boxes that differ only in their power management then show the same frac_at_measured_clock."""
    out = dict(roof, **clocks)
    if clocks.get("sclk_mhz_mean"):
        out["frac_at_measured_clock"] = roof["frac"] * NOMINAL_SCLK_MHZ / clocks["sclk_mhz_mean"]
    return out


# ---------------------------------------------------------------------------------------------------------------
# synthetic workloads; tests/test_bench_inputs_gpu.py runs the oracle on exactly these
# ---------------------------------------------------------------------------------------------------------------
def synth_lmm_inputs(N, seed, device):
    """C3: K = G G^T from lineage-structured binary markers, scaled N/trace; D = 1; binary phenotype with a heritable
    component; decomposition + h2 by the package's own initialise_lmm restatement."""
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
    # (SEERHIP_BENCH_CPU_EIGH=1: numpy's eigensolver -- rocSOLVER's segfaults under rocprofv3 counter collection, tools/profile_r03.sh)
    U, S, h2, nll, C = initialise_lmm_arrays(K, y, None, use_gpu=not os.environ.get("SEERHIP_BENCH_CPU_EIGH"))
    return U, S, h2, C, y, lin


def synth_glm_inputs(N, q, seed=1002):
    """C2 / C4: MDS-like covariates scaled by their max-abs (pyseer/input.py:135-136), Bernoulli phenotype, null fits."""
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.5 * W[:, 0] - W[:, 1]
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-eta))).astype(np.float64)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf
    nf = fit_null(y, W, e0, False, firth=True)
    return y, W, nl, nf


def synth_bits(V, N, row_bytes, seed, device, chunk=1 << 16):
    """Packed presence rows, variant-major, generated on the device: AF ~ U(0.02, 0.98) (98 %), rare (2 %), i.i.d.
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


def unpack_rows(bits_u8, N):
    """(V, row_bytes) uint8 host array -> (V, N) float64 0/1 (LSB-first, as pyseer_amd/packing.py packs)."""
    return np.unpackbits(np.ascontiguousarray(bits_u8), axis=1, bitorder="little")[:, :N].astype(np.float64)


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (oracle/ = test infrastructure: only this leg, tests/ and smoke() may touch it), bounded samples
# ---------------------------------------------------------------------------------------------------------------
def effective_cpus():
    """CPUs this process can actually use: the affinity mask, cut by the cgroup CPU quota (the GPU boxes show 256 logical CPUs under a
    quota of 16: more busy threads than that only time-slice, and spinning ones make everything slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(np.ceil(float(q) / float(per)))))
    except (IOError, OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (IOError, OSError, ValueError):
            pass
    return n


def cpu_baseline_lmm(U, S, y, C, h2, N, block=1000, blocks_per_proc=3):
    """The reference's own formulation on this host's cores, run the way `pyseer --cpu P` runs it: P worker processes, one BLAS thread
    each, whole blocks of variants per task; per block: residualise, `U.T.dot(A)` through BLAS (lmm_cov.py:186), quadratic forms, F tail
    (oracle/lmm_blas.py).  A separate interpreter forks the workers (oracle/cpu_baseline_lmm.py), so nothing GPU-side is forked."""
    import subprocess
    import tempfile
    procs = max(1, min(effective_cpus(), (os.cpu_count() or 2) // 2))    # usable cores: cgroup quota, and no SMT siblings (nothing for dgemm)
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "lmm_inputs.npz")
        np.savez(f, U=U, S=S, y=y, C=C, h2=h2)
        env = dict(os.environ); env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        out = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline_lmm", f, str(procs), str(block), str(blocks_per_proc)],
                             cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + out.stderr[-2000:])
    r = json.loads(out.stdout.strip().splitlines()[-1])
    return dict(value=r["variants"] / r["seconds"], unit="variants/s", cores=r["procs"], kind="port", host_logical_cpus=os.cpu_count(),
                sample="%d synthetic k-mers x %d samples: %d worker processes (as pyseer --cpu; = the CPUs this container may use), one BLAS thread each, %d blocks of %d variants "
                       "per worker; fit_lmm_block restated with numpy (oracle/lmm_blas.py): U.T.dot(A) through OpenBLAS dgemm as the reference "
                       "issues it, rows drawn before the clock starts; %.1f variants/s per core.  The reference itself (fit_lmm_block through "
                       "numpy, SURVEY.md 3.3): %.0f variants/s on one thread" % (r["variants"], N, r["procs"], blocks_per_proc, r["block"],
                                                               r["variants"] / r["seconds"] / r["procs"], REFERENCE_PER_CORE["lmm"]),
                reference_per_core=REFERENCE_PER_CORE["lmm"])


def cpu_baseline_glm(y, W, nl, nf, N, force_firth, target_s=12.0):
    """oracle/seer_oracle.c (C port of model.py:202-504, OpenMP over variants) on a bounded sample of the same workload."""
    from oracle import oracle as orc
    ncores = effective_cpus()
    orc.set_threads(ncores)
    rng = np.random.default_rng(78)

    def run(v):
        af = rng.uniform(0.02, 0.98, v)
        Kv = (rng.random((v, N)) < af[:, None]).astype(np.float64)
        t0 = time.time()
        if force_firth:
            orc.firth_batch(y, Kv, W)
        else:
            orc.fixed_effects_batch(y, Kv, W, False, 1.0, 1.0, nl, nf)
        return time.time() - t0
    v0 = 2 * ncores
    t = run(v0)
    per = int(min(max(v0, v0 * 3.0 / max(t, 1e-3)), (1 << 30) // (8 * N)))   # ~3 s per call, at most 1 GB of rows
    per = max(ncores, per // ncores * ncores)
    done, spent = 0, 0.0
    while spent < target_s:
        spent += run(per); done += per
    what = "fit_firth on every variant (orc_firth_batch)" if force_firth else "fixed_effects_regression (orc_fixed_effects_batch)"
    refpc = REFERENCE_PER_CORE["firth" if force_firth else "logistic"]
    return dict(value=done / spent, unit="variants/s", cores=ncores, kind="port",
                host_logical_cpus=os.cpu_count(),
                sample="%d synthetic k-mers x %d samples, %d covariates, %s, oracle/seer_oracle.c with OpenMP on %d threads (= the CPUs this "
                       "container may use): a C port, %.0fx faster per core than the reference itself, which runs %s variants/s on one thread "
                       "(statsmodels / numpy, SURVEY.md 3.3)" % (done, N, W.shape[1], what, ncores, (done / spent / ncores) / refpc, refpc),
                reference_per_core=refpc)


# ---------------------------------------------------------------------------------------------------------------
# committed counter measurements (PMC passes cannot run inside this process: they need their own rocprofv3 runs)
# ---------------------------------------------------------------------------------------------------------------
def _profile_json(name):
    for d in PROFILE_DIRS:
        p = os.path.join(ROOT, "profiles", d, name)
        if os.path.exists(p):
            return json.load(open(p)), "profiles/%s/%s" % (d, name)
    return None, None


def measured_traffic(tag, Vs):
    """HBM-side bytes per launch of the dominant kernel(s) from the committed FETCH_SIZE / WRITE_SIZE passes, scaled to the launch."""
    t, src = _profile_json("traffic_%s.json" % tag)
    if t is None:
        return None, None
    per_variant = (t["FETCH_SIZE_KB"] * t["fetch_correction"] + (t.get("WRITE_SIZE_KB") or 0.0)) * 1024.0 / t["variants_per_dispatch"]
    return per_variant * Vs, "%s: %s" % (src, t["source"])


def measured_flops(tag):
    """fp64/fp32 lane-flops per variant of the fixed-effects kernels, counted by SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64/F32 and
    SQ_INSTS_VALU_MFMA_MOPS_F32 over one batch of the same synthetic workload (tools/profile_r02.sh)."""
    t, src = _profile_json("flops_%s.json" % tag)
    if t is None:
        return None, None, None, None, None
    vi = t.get("counters", {}).get("SQ_INSTS_VALU")
    return (t["fp64_flops_per_variant"], t.get("fp32_flops_per_variant", 0.0), "%s: %s" % (src, t["source"]), t.get("fp16_mfma_flops_per_variant", 0.0),
            None if not vi else vi / t["variants_per_step"])


# ---------------------------------------------------------------------------------------------------------------
def glm_roofline(cfg, q, Vs, kern_s, klaunch, rb):
    """roofline object of a fixed-effects configuration: fp64 lane-flops counted by PMC over one batch (profiles/rNN/flops_<cfg>.json) x variants
    per step / the HIP-event time of the fit's kernels per step, against the fp64 vector peak; traffic from the committed FETCH/WRITE passes."""
    force = cfg == "C4"
    tag = cfg.lower()
    f64, f32, fsrc, f16, vinst = measured_flops(tag)
    traffic, traffic_src = measured_traffic(tag, Vs)
    achieved = None if f64 is None else f64 * Vs / kern_s / 1e12
    return {"bound": "valu", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": None if achieved is None else achieved / FP64_VECTOR_PEAK_TFLOPS,
            "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
            "kernel": GLM_KERNELS[force], "kernel_ms": kern_s * 1e3, "launches": klaunch,
            "ops": "fp64 lane-flops actually executed per variant (PMC: 64 x (2 FMA_F64 + ADD_F64 + MUL_F64 + TRANS_F64)) x variants per step "
                   "/ HIP-event time of those kernels per step; single-precision work (first Newton rounds, the first Firth pass) and the f16 MFMA work are reported beside it, not "
                   "added.  This fraction falls when a round removes fp64 arithmetic (C4: 4.76 M fp64 flop per variant in round 3, 1.03 M now, at 1.8x the "
                   "variants/s): valu_issue_frac is the one that says how busy the binding unit is",
            "fp64_flops_per_variant": f64, "fp32_flops_per_variant": f32, "fp16_mfma_flops_per_variant": f16, "flops_source": fsrc,
            "mfma_f16_tflops": None if not f16 else f16 * Vs / kern_s / 1e12,
            # what binds these kernels is the vector ALU's ISSUE rate, whatever the precision (fp64, packed fp32 and conversions each take one
            # 4-cycle slot per wavefront instruction; tools/ubench/valu_rate.hip): wavefront-level VALU instructions per variant (PMC SQ_INSTS_VALU,
            # MFMAs included) x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz)
            "valu_wave_insts_per_variant": vinst,
            "valu_issue_frac": None if not vinst else vinst * Vs * 4.0 / (1024.0 * kern_s * NOMINAL_SCLK_MHZ * 1e6),
            "hbm_algorithmic_GBps": (rb + (5 + q) * 8 + 4) * Vs / kern_s / 1e9}


def glm_metric(cfg, N):
    force = cfg == "C4"
    return ("k-mer tests/sec at N=%d samples (fixed effects: %s), whole job" % (N, "Firth" if force else "logistic"),
            "f64 (eta, likelihood, score, solves) + f16 hi/lo MFMA with fp32 accumulation (information matrix, third-moment tensor of the penalty); f32 for the first two of the three passes (round 6: sums in f64 across groups; the second is discarded where a fit contracts slowly or its kbeta is small against what it leaves); f64 throughout for the fits the exact kernels take" if force
            else "f64 (eta, score, likelihood, the last Newton step, solves); the final information matrix = the null model's matrix and the carrier sums in exact f64 / int8 limbs + the DIFFERENCES w - w0 summed in f32 (f16 hi/lo MFMA products, fp32 accumulation; the variant's row per 64-sample word in f32, across words in f64); f32 / f16-MFMA Hessians in the Newton steering passes")


def glm_workload(cfg, Vs, N, q):
    what = "Firth-penalised logistic regression on every variant (force_firth)" if cfg == "C4" else "logistic regression (Firth for routed variants)"
    return {"workload": "%s: %s, %d synthetic k-mers x %d samples per step per GPU, %d covariates, inputs resident in HBM" % (cfg, what, Vs, N, q),
            "variants_per_step_per_gpu": Vs, "n_samples": N, "sharding": "k-mer stream sharded by rank, no collective"}


def glm_lanes_line(eng, cfg, bits, q, Vs, rb, steps=15, warmup=None, check=None):
    """The fixed-effects steps through the context's LANES (sh_glm_batch_dev_async, csrc/lanes_api.inc; the library's default for a job):
    a step is still one batch through one sh_glm_batch_dev -- on one of the context's worker threads, each with its own stream and
    workspaces -- and the timed region runs exactly `steps` of them, submitted back to back and completed by sh_wait, bracketed by device
    synchronisation.  One stream of batches leaves the device idle between its per-variant kernels, behind the list-length read-backs of
    its launch code and in the tails of its passes; batches of consecutive calls fill that.  (The counterpart of the reference's pool of
    --cpu N workers over blocks of variants, pyseer/__main__.py:541-568.)  Every step has its own result buffers.
    check = (index into bits, out, fl) of a synchronous call: the lanes' rows for those bits must be the same bytes."""
    import torch
    dev = bits[0].device
    lanes = eng.get_lanes()
    warmup = 2 * lanes if warmup is None else warmup
    nb = len(bits)
    nres = max(steps, 1)                                               # every timed step its own result buffers (lanes complete out of order)
    outs = [torch.empty((5 + q, Vs), dtype=torch.float64, device=dev) for _ in range(nres)]
    fls = [torch.empty((Vs,), dtype=torch.int32, device=dev) for _ in range(nres)]
    for i in range(warmup):
        eng.glm_batch_dev_async(bits[i % nb], outs[i % nres], fls[i % nres])
    eng.wait(); torch.cuda.synchronize()
    eng.set_timing(True)
    sampler = ClockSampler(dev.index or 0).start()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.glm_batch_dev_async(bits[i % nb], outs[i % nres], fls[i % nres])
    eng.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    clocks = sampler.stop()
    span_ms, nspan = eng.get_timing()
    eng.set_timing(False)
    step_s = dt / steps
    roof = with_clock(glm_roofline(cfg, q, Vs, step_s, steps, rb), clocks)
    roof["kernel_ms_is"] = ("timed region / steps (device synchronised on both sides): the batches of different lanes overlap on the device, so a batch's own "
                            "HIP-event span (batch_span_ms, on its lane's stream) is longer than the time the device spends per step")
    roof["batch_span_ms"] = span_ms / max(nspan, 1)
    res = {"lanes": lanes, "contexts": 1, "value": Vs * steps / dt, "unit": "variants/s", "steps": steps, "warmup": warmup, "ms_per_step": step_s * 1e3,
           "roofline": roof,
           "what": "ONE engine context, batches through its %d lanes (sh_glm_batch_dev_async + sh_wait): worker threads with their own stream and "
                   "workspaces inside the library; one step = one batch through one sh_glm_batch_dev on a lane" % lanes}
    last = (steps - 1) % nres
    res["_last"] = (bits[(steps - 1) % nb], outs[last], fls[last])
    if check is not None:
        bi, out0, fl0 = check
        j = None
        for i in range(steps - 1, max(steps - 1 - nres, -1), -1):     # a step of the timed region that ran those rows and whose buffers were not reused since
            if i % nb == bi:
                j = i % nres; break
        if j is None:
            eng.glm_batch_dev_async(bits[bi], outs[0], fls[0]); eng.wait(); torch.cuda.synchronize(); j = 0
        same = bool(torch.equal(outs[j].view(torch.int64), out0.view(torch.int64))) and bool(torch.equal(fls[j], fl0))
        res["identical_to_synchronous_call"] = same
        if not same:                                                  # (never seen; reported rather than fatal)
            a, b = outs[j].double(), out0.double()
            okm = torch.isfinite(a) & torch.isfinite(b)
            res["flags_identical"] = bool(torch.equal(fls[j], fl0))
            res["max_rel_dev_from_synchronous_call"] = float(((a - b).abs() / b.abs().clamp_min(1e-300))[okm].max().item()) if bool(okm.any()) else None
    return res


def fixed_effects_line(cfg, dev, local, steps=5, warmup=1, Vs=None, cpu=True, parity=True, env=None):
    """One fixed-effects configuration measured the way the main line is (HIP events on the launch stream, inputs resident, oracle re-check
    of the timed output, CPU baseline): the `extra` entries of the default (C3) line, so that the driver's run of `bench.py --gpus 1`
    carries both halves of BASELINE's metric (LMM and fixed effects at N = 5000).  env: switches read by sh_glm_setup (e.g. the literal
    Firth step-halving rule)."""
    import torch
    from pyseer_amd.engine import Engine, row_bytes_for
    N = 1000 if cfg == "C2" else N_SAMPLES
    q = 10
    Vs = Vs or ((1 << 20) if cfg == "C2" else (1 << 18))
    rb = row_bytes_for(N)
    y, W, nl, nf = synth_glm_inputs(N, q)
    saved = {}
    for k, v in (env or {}).items():
        saved[k] = os.environ.get(k); os.environ[k] = v
    try:
        eng = Engine(N, device=local); eng.use_torch_stream(); eng.set_af_filter(0.01, 0.99)
        eng.glm_setup(y, W, False, nl, nf, 1.0, 1.0, force_firth=(cfg == "C4"))
    finally:
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    nb = min(steps + warmup, 4)
    bits = [synth_bits(Vs, N, rb, 4242 + i, dev) for i in range(nb)]
    out = torch.empty((5 + q, Vs), dtype=torch.float64, device=dev); fl = torch.empty((Vs,), dtype=torch.int32, device=dev)
    for i in range(warmup):
        eng.glm_batch_dev(bits[i % nb], out, fl)
    torch.cuda.synchronize()
    eng.set_timing(True)
    sampler = ClockSampler(local).start()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.glm_batch_dev(bits[(warmup + i) % nb], out, fl)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    clocks = sampler.stop()
    kms, klaunch = eng.get_timing()
    eng.set_timing(False)
    kern_s = kms / max(klaunch, 1) * 1e-3
    metric, dtype = glm_metric(cfg, N)
    sync = {"value": Vs * steps / dt, "unit": "variants/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "roofline": with_clock(glm_roofline(cfg, q, Vs, kern_s, klaunch, rb), clocks),
            "what": "the same context, one synchronous sh_glm_batch_dev per step on one stream (rounds 1-4's measurement)"}
    if env:
        # a switch read at set-up (the literal step-halving rule): measured the old way only
        res = {"metric": metric, "dtype": dtype, "config": glm_workload(cfg, Vs, N, q), **sync, "env": dict(env), "lanes": 1}
        last = (bits[(warmup + steps - 1) % nb], out, fl)
    else:
        ln = glm_lanes_line(eng, cfg, bits, q, Vs, rb, check=((warmup + steps - 1) % nb, out, fl))
        last = ln.pop("_last")
        res = {"metric": metric, "dtype": dtype, "config": glm_workload(cfg, Vs, N, q), **ln, "synchronous": sync}
        res["config"]["workload"] += ", one engine context, batches through its %d lanes" % ln["lanes"]
    if parity:
        n, devs = parity_glm(y, W, nl, nf, cfg == "C4", last[0], last[1], last[2], N)
        res["parity_checked"] = n; res["parity_max_rel_dev"] = devs
    eng.close()
    del last
    del bits, out, fl
    if cpu:
        res["cpu_baseline"] = cpu_baseline_glm(y, W, nl, nf, N, cfg == "C4")
    return res


def lmm_variant_line(U, S, y, C, h2, dev, local, bits, steps, limbs=0, tol=None):
    """The LMM step again with another limb count or refinement tolerance (tol = 1e-300: every variant's bound exceeds it, so every variant is
    re-contracted with the extra limbs = 56 bits in all, the fp64-equivalent throughput; a tolerance of exactly 0 means "never" in the ABI), on
    the main line's own rows; oracle re-check of its timed output."""
    import torch
    from pyseer_amd.engine import Engine
    N = U.shape[0]
    eng = Engine(N, device=local); eng.use_torch_stream()
    eng.lmm_setup(U, S, y, C, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0, n_limbs=limbs)
    if tol is not None:
        eng.set_lmm_tol(tol)
    Vs = bits[0].shape[0]
    out = torch.empty((5, Vs), dtype=torch.float64, device=dev); fl = torch.empty((Vs,), dtype=torch.int32, device=dev)
    eng.lmm_batch_dev(bits[0], out, fl); torch.cuda.synchronize()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for i in range(steps):
        eng.lmm_batch_dev(bits[i % len(bits)], out, fl)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms, klaunch = eng.get_timing()
    info = eng.lmm_info()
    kern_s = kms / max(klaunch, 1) * 1e-3
    n, devs = parity_lmm(U, S, y, C, h2, bits[(steps - 1) % len(bits)], out, N)
    res = {"value": Vs * steps / dt, "unit": "variants/s", "ms_per_step": dt / steps * 1e3, "n_limbs": info["n_limbs"], "refine_tol": info.get("refine_tol"),
           "kernel_ms": kern_s * 1e3, "frac_of_int8_peak_main_pass": 2.0 * info["int8_macs_per_variant"] * Vs / kern_s / 1e12 / INT8_DENSE_PEAK_TOPS,
           "refined_last_batch": info.get("refined_last_batch"), "parity_checked": n, "parity_max_rel_dev": devs}
    eng.close()
    return res


def lmm_host_pointer_line(U, S, y, C, h2, dev, local, bits_t, steps):
    """The LMM step through the HOST-pointer entry point (sh_lmm_batch_async: pageable numpy rows in, numpy results out, PCIe both ways,
    calls pipelined as the command line issues them).  Reported beside the line, never as its `value` (inputs resident in HBM)."""
    import torch
    from pyseer_amd.engine import Engine
    N = U.shape[0]
    eng = Engine(N, device=local)
    eng.lmm_setup(U, S, y, C, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0)
    host = bits_t.cpu().numpy()
    Vs = host.shape[0]
    want = eng.lmm_batch(host)                                        # warm-up (staging buffers), and the synchronous call's result
    t0 = time.perf_counter(); eng.lmm_batch(host); dt_sync = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = [eng.lmm_batch(host, pipelined=True) for _ in range(steps)]
    eng.wait()
    dt = time.perf_counter() - t0
    same = all(np.array_equal(want[k], g[k], equal_nan=True) for g in got for k in want)
    eng.close()
    return {"value": Vs * steps / dt, "unit": "variants/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "synchronous_call_ms": dt_sync * 1e3,
            "bytes_per_variant_over_pcie": {"in": int(host.shape[1]), "out": 44}, "identical_to_synchronous_call": bool(same),
            "what": "sh_lmm_batch_async on host memory (pageable numpy rows -> pinned staging -> HBM, results back to numpy), %d back-to-back "
                    "calls of %d variants + sh_wait; PCIe-inclusive, not the line's value" % (steps, Vs)}


def rel_dev(got, want, floor=1e-300):
    got = np.asarray(got, dtype=float); want = np.asarray(want, dtype=float)
    ok = np.isfinite(want) & np.isfinite(got)
    if not ok.any():
        return 0.0
    return float(np.max(np.abs(got[ok] - want[ok]) / np.maximum(np.abs(want[ok]), floor)))


def parity_lmm(U, S, y, C, h2, bits_t, out_t, N, n_check=64):
    """Re-check n_check variants of the timed output against the CPU oracle (orc_lmm_block: the reference's fit_lmm_block)."""
    from oracle import oracle as orc
    orc.set_threads(effective_cpus())
    V = bits_t.shape[0]
    idx = np.linspace(0, V - 1, n_check).astype(np.int64)
    import torch
    ti = torch.from_numpy(idx).to(bits_t.device)
    rows = unpack_rows(bits_t[ti].cpu().numpy(), N)
    got = out_t[:, ti].cpu().numpy()                                     # prep, p, beta, bse, frac_h2
    wb, ws, wf, wp = orc.LmmOracle(U, S, y, C).block(h2, rows)
    dev = {"beta": rel_dev(got[2], wb), "bse": rel_dev(got[3], ws), "frac_h2": rel_dev(got[4], wf, 1e-12), "pvalue": rel_dev(got[1], wp)}
    return len(idx), dev


def parity_glm(y, W, nl, nf, force_firth, bits_t, out_t, fl_t, N, n_check=64):
    from oracle import oracle as orc
    import torch
    orc.set_threads(effective_cpus())
    V = bits_t.shape[0]
    idx = np.linspace(0, V - 1, n_check).astype(np.int64)
    ti = torch.from_numpy(idx).to(bits_t.device)
    rows = unpack_rows(bits_t[ti].cpu().numpy(), N)
    got = out_t[:, ti].cpu().numpy(); fl = fl_t[ti].cpu().numpy().astype(np.uint32)
    af = rows.mean(axis=1); inwin = (af >= 0.01) & (af <= 0.99)           # the engine's AF window in this bench
    if force_firth:
        w = orc.firth_batch(y, rows, W)
        ok = inwin & (w["status"] == 0)
        lr = -2.0 * (nf - w["fitll"])
        wp = np.array([orc.chi2_sf1(x) if x > 0 else 1.0 for x in lr])
        # coefficients: |d| / max(|want|, 0.02), i.e. the tests' tolerance for Firth rows (1e-6 relative or 2e-8 absolute: fit_firth itself stops a
        # step of ~1e-8 short of its fixed point, and a k-mer without effect has |kbeta| ~ 1e-4); the largest absolute deviation beside it
        dev = {"kbeta": rel_dev(got[2][ok], w["kbeta"][ok], 0.02), "bse": rel_dev(got[3][ok], w["bse"][ok]),
               "intercept": rel_dev(got[4][ok], w["intercept"][ok], 0.02), "pvalue": rel_dev(got[1][ok], wp[ok]),
               "kbeta_abs": float(np.max(np.abs(got[2][ok] - w["kbeta"][ok]))) if ok.any() else 0.0}
    else:
        w = orc.fixed_effects_batch(y, rows, W, False, 1.0, 1.0, nl, nf)
        ok = inwin
        dev = {f: rel_dev(got[i][ok], w[f][ok]) for i, f in ((0, "prep"), (1, "pvalue"), (2, "kbeta"), (3, "bse"), (4, "intercept"))}
        dev["notes_equal"] = bool(((fl[ok] & 0x1FF) == w["notes"][ok]).all())
    return int(ok.sum()), dev


# ---------------------------------------------------------------------------------------------------------------
# what goes on the ONE JSON line.  Round 4's line was 19 KB (prose in `what` / `ops` / `sample` / `workload`), and the driver keeps the parsed
# line plus an 8 KB tail: the fixed-effects half of the metric fell off both.  The line now carries numbers; the prose goes to stderr
# ("bench details ...", printed BEFORE the line) and to gpurun_out/bench_details.json.
# ---------------------------------------------------------------------------------------------------------------
ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "launches", "valu_issue_frac", "frac_at_measured_clock",
             "sclk_mhz_mean", "power_w_mean", "fp64_equiv_tflops", "hbm_algorithmic_GBps", "mfma_f16_tflops", "int8_macs_per_variant", "algorithmic_bytes_per_launch")


def max_dev(d):
    v = [float(x) for k, x in d.items() if isinstance(x, (int, float)) and not isinstance(x, bool) and not k.endswith("_abs")]
    return max(v) if v else None


def rounded(o, sig=6):
    """Floats of a JSON-able object to `sig` significant digits (the line is for reading; full precision is in the details)."""
    if isinstance(o, float):
        return float("%.*g" % (sig, o)) if o == o and abs(o) != float("inf") else o
    if isinstance(o, dict):
        return {k: rounded(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [rounded(v, sig) for v in o]
    return o


def slim(e):
    """A secondary measurement reduced to its numbers."""
    if not isinstance(e, dict):
        return e
    out = {k: e[k] for k in ("value", "unit", "steps", "ms_per_step", "n_limbs", "kernel_ms", "frac_of_int8_peak_main_pass", "refined_last_batch",
                             "parity_checked", "bytes_per_variant_over_pcie", "identical_to_synchronous_call", "contexts", "lanes",
                             "valu_issue_frac", "env", "error") if k in e}
    if isinstance(e.get("parity_max_rel_dev"), dict) and e["parity_max_rel_dev"]:
        out["parity_max_rel_dev"] = max_dev(e["parity_max_rel_dev"])
    if isinstance(e.get("roofline"), dict):
        out["roofline"] = {k: e["roofline"][k] for k in ROOF_KEYS if k in e["roofline"] and k != "kernel"}
    if isinstance(e.get("cpu_baseline"), dict):
        out["cpu_baseline"] = {k: e["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind") if k in e["cpu_baseline"]}
    for k in ("synchronous", "pipelined"):
        if isinstance(e.get(k), dict):
            out[k] = slim(e[k])
    if isinstance(e.get("config"), dict) and "workload" in e["config"]:
        out["workload"] = e["config"]["workload"].split(":")[0]
    return out


def fixed_effects_summary(extra):
    """BASELINE's metric is "LMM and fixed-effects at N = 5000": the fixed-effects half as top-level numbers of the line."""
    def one(e):
        if not isinstance(e, dict) or "value" not in e:
            return None
        r = e.get("roofline", {})
        o = {"value": e["value"], "unit": e.get("unit"), "ms_per_step": e.get("ms_per_step"), "frac": r.get("frac"), "valu_issue_frac": r.get("valu_issue_frac"),
             "bound": r.get("bound"), "traffic": r.get("traffic"), "contexts": 1, "lanes": e.get("lanes", 1)}
        if isinstance(e.get("parity_max_rel_dev"), dict) and e["parity_max_rel_dev"]:
            o["parity_max_rel_dev"] = max_dev(e["parity_max_rel_dev"])
        t = e.get("synchronous")
        if isinstance(t, dict) and "value" in t:
            o["synchronous_value"] = t["value"]; o["synchronous_valu_issue_frac"] = t.get("roofline", {}).get("valu_issue_frac")
        return o
    return {"logistic": one(extra.get("C2N5000")), "firth": one(extra.get("C4")),
            "workloads": "C2N5000 / C4: 2^18 synthetic k-mers x 5000 samples per step, 10 covariates, rows resident in HBM, ONE engine context, "
                         "batches through its lanes (sh_glm_batch_dev_async); synchronous_value = one sh_glm_batch_dev per step on one stream"}


def emit(res):
    """Details to stderr and gpurun_out/, then the compact line (the last thing on stdout)."""
    line = dict(res)
    if isinstance(res.get("extra"), dict):
        line["fixed_effects_n5000"] = fixed_effects_summary(res["extra"])
        line["extra"] = {k: slim(v) for k, v in res["extra"].items()}
    if isinstance(res.get("roofline"), dict):
        line["roofline"] = {k: v for k, v in res["roofline"].items() if k not in ("ops", "traffic_source", "traffic_unit", "clock_source")}
    if isinstance(res.get("cpu_baseline"), dict) and len(json.dumps(res["cpu_baseline"])) > 400:
        cb = dict(res["cpu_baseline"]); cb["sample"] = str(cb.get("sample", ""))[:160]; line["cpu_baseline"] = cb
    if isinstance(res.get("synchronous"), dict):
        line["synchronous"] = slim(res["synchronous"])
    order = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
             "fixed_effects_n5000", "roofline", "cpu_baseline"]
    line = {**{k: line[k] for k in order if k in line}, **{k: v for k, v in line.items() if k not in order}}
    details = json.dumps(res)
    sys.stderr.write("bench details " + details + "\n"); sys.stderr.flush()
    try:
        o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(o, exist_ok=True)
        with open(os.path.join(o, "bench_details.json"), "w") as f:
            f.write(details + "\n")
    except OSError:
        pass
    print(json.dumps(rounded(line))); sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C3", "C2", "C2N5000", "C4"])
    ap.add_argument("--variants-per-step", type=int, default=0, help="default: 2^20 (C3, C2), 2^18 (C2N5000, C4)")
    ap.add_argument("--limbs", type=int, default=0, help="int8 limbs of the LMM contraction (0 = the library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary fixed-effects measurements of the C3 line")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle re-check of the timed output")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for the one-GPU test)")
    ap.add_argument("--one-device", action="store_true", help="test only: every rank uses cuda:0 (exercises the N > 1 code path on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # how many ranks the collective library really connected (the driver's "did RCCL see N ranks" check): an all-reduce of ones
    from pyseer_amd import parallel as par
    cdev = dev if args.backend == "nccl" else "cpu"              # where the collectives' tensors live
    ranks_seen = par.ranks_connected(cdev)
    if world > 1:
        assert ranks_seen == world == dist.get_world_size(), (ranks_seen, world)

    from pyseer_amd.engine import Engine, row_bytes_for
    cfg = args.config
    lmm = cfg == "C3"
    N = 1000 if cfg == "C2" else N_SAMPLES
    q = 10
    Vs = args.variants_per_step or ((1 << 20) if cfg in ("C3", "C2") else (1 << 18))
    rb = row_bytes_for(N)
    eng = Engine(N, device=local)
    eng.use_torch_stream()

    if lmm:
        # per-run constants: rank 0 decomposes the kinship once and broadcasts (U, S, h2, y) at set-up time; the data path has no collective
        consts = None
        if rank == 0:
            U, S, h2, C, y, _ = synth_lmm_inputs(N, 1003, dev)
            consts = {"U": U, "S": S, "y": y, "h2": np.array([h2]), "C": C}
        if world > 1:
            consts = par.broadcast_run_constants(consts, 0, cdev)
            U, S, y, h2, C = consts["U"], consts["S"], consts["y"], float(consts["h2"][0]), consts["C"]
        eng.lmm_setup(U, S, y, C, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0, n_limbs=args.limbs)
        info = eng.lmm_info()
        nrow = 5
        run = eng.lmm_batch_dev
    else:
        y, W, nl, nf = synth_glm_inputs(N, q)
        eng.set_af_filter(0.01, 0.99)
        eng.glm_setup(y, W, False, nl, nf, 1.0, 1.0, force_firth=(cfg == "C4"))
        nrow = 5 + q
        run = None                                               # (the lanes: below)

    nbuf = args.steps + args.warmup
    # every step has its own rows (seed + rank shard), all resident in HBM before the timed region; at most 16 distinct buffers (10 GB)
    ndist = min(nbuf, 16)
    bits = [synth_bits(Vs, N, rb, 1003 + 1000 * rank + i, dev) for i in range(ndist)]
    if lmm:
        out = torch.empty((nrow, Vs), dtype=torch.float64, device=dev)
        fl = torch.empty((Vs,), dtype=torch.int32, device=dev)
        step = lambda i: run(bits[i % ndist], out, fl)
        finish = lambda: None
    else:
        # fixed effects: every step is one batch through one sh_glm_batch_dev on one of the context's lanes (sh_glm_batch_dev_async: worker
        # threads with their own stream and workspaces inside the library, csrc/lanes_api.inc); the timed region submits exactly K of them
        # and completes them (sh_wait) before the closing synchronisation.  Every batch in flight has its own result buffers.
        lanes = eng.get_lanes()
        nres = nbuf                                              # (lanes complete out of order: a buffer is never handed out twice)
        outs = [torch.empty((nrow, Vs), dtype=torch.float64, device=dev) for _ in range(nres)]
        fls = [torch.empty((Vs,), dtype=torch.int32, device=dev) for _ in range(nres)]
        step = lambda i: eng.glm_batch_dev_async(bits[i % ndist], outs[i % nres], fls[i % nres])
        finish = eng.wait
        out, fl = outs[(nbuf - 1) % nres], fls[(nbuf - 1) % nres]   # the last timed step's
    for i in range(args.warmup):
        step(i)
    finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    eng.set_timing(True)
    sampler = ClockSampler(local).start() if rank == 0 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt_local = time.perf_counter() - t0
    clocks = sampler.stop() if sampler is not None else {}
    dt = dt_local
    kms, klaunch = eng.get_timing()
    if not lmm and ("glm" in os.environ.get("SEERHIP_DEBUG", "").split(",")):
        eng.glm_info()                                       # the library prints its development counters to stderr
    # sanity: the timed work produced finite statistics (row 2 = beta / kbeta; AF-filtered rows of the fixed-effects configs are NaN by contract)
    fin = torch.isfinite(out[2])
    if not lmm:
        # NaN by contract: rows outside the AF window (note bit 0) and rows whose fit_firth returned None (firth-fail, bit 6: with the reference's
        # literal step-halving rule a few 1e-4 of forced-Firth fits exhaust step_limit on a last-bit tie)
        fin = fin | ((fl & 1) != 0) | ((fl & 64) != 0)
    frac_finite = float(fin.double().mean().item())
    per_rank = [float(Vs) * args.steps / dt_local]
    fin_all = [frac_finite]
    if world > 1:
        dt = par.max_over_ranks(dt, cdev)
        g = par.gather_floats([per_rank[0], frac_finite], cdev)
        per_rank = [x[0] for x in g]; fin_all = [x[1] for x in g]
    assert min(fin_all) == 1.0, "a rank produced non-finite statistics: %s" % fin_all

    if rank == 0:
        last = bits[(args.warmup + args.steps - 1) % ndist]
        total = float(Vs) * args.steps * world
        value = total / dt
        # LMM: the HIP-event time of the contraction kernel per launch.  Fixed effects on the lanes: the batches of different lanes overlap on
        # the device, so the time the device spends per step is the timed region / steps (a batch's own event span is reported beside it)
        kern_s = kms / max(klaunch, 1) * 1e-3 if lmm else dt_local / args.steps
        res = {"value": value, "unit": "variants/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "data": "synthetic", "world": world, "rccl_ranks_seen": ranks_seen, "collective_backend": args.backend if world > 1 else None,
               "per_rank_value": per_rank, "finite_fraction": min(fin_all),
               "distinct_rows_per_gpu": int(ndist) * int(Vs)}
        if lmm:
            info = eng.lmm_info()                         # after the timed launches: the MACs the contraction kernel actually issued per variant
            int8_ops = 2.0 * info["int8_macs_per_variant"] * Vs
            achieved = int8_ops / kern_s / 1e12
            traffic, traffic_src = measured_traffic("lmm", Vs)
            res.update({
                "metric": "k-mer tests/sec at N=5000 samples (LMM), whole job",
                "dtype": "int8 limbs x%d (exact int32 MFMA accumulation, fp64 recombination and statistics)" % info["n_limbs"],
                "config": {"workload": "C3: LMM (FaST-LMM per-variant test), %d synthetic k-mers x %d samples per step per GPU, "
                                       "D=1, k=%d, h2=%.4f, inputs resident in HBM" % (Vs, N, U.shape[1], h2),
                           "variants_per_step_per_gpu": Vs, "n_samples": N, "sharding": "k-mer stream sharded by rank, no collective"},
                "roofline": with_clock({"bound": "mfma", "achieved": achieved, "peak": INT8_DENSE_PEAK_TOPS, "unit": "TFLOP/s",
                             "frac": achieved / INT8_DENSE_PEAK_TOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                             "traffic_source": traffic_src,
                             "kernel": "k_lmm_quadform_i8w", "kernel_ms": kern_s * 1e3, "launches": klaunch,
                             "ops": "int8 multiply-adds x2 actually issued per variant (sh_lmm_info) x variants per launch",
                             "int8_macs_per_variant": info["int8_macs_per_variant"],
                             "fp64_equiv_tflops": FP64_FLOP_PER_TEST * Vs / kern_s / 1e12,
                             "hbm_algorithmic_GBps": ALGO_BYTES_PER_TEST * Vs / kern_s / 1e9}, clocks),
                "error_bound": {k: info[k] for k in ("quant_err_norm", "quant_err_norm_power_iteration", "quant_err_norm_squarings", "bound_rel_typical",
                                                     "refined_last_batch") if k in info},
            })
            if not args.no_parity:
                n, devs = parity_lmm(U, S, y, C, h2, last, out, N)
                res["parity_checked"] = n; res["parity_max_rel_dev"] = devs
                assert max(devs.values()) < 1e-6, "timed output deviates from the oracle: %s" % devs
            if world == 1 and not args.no_extra:
                # the rest of BASELINE's metric ("LMM and fixed-effects at N = 5000") on the same line: each entry is measured like a main line
                # (event-timed kernels, counted roofline, oracle re-check, CPU baseline).  C4_literal: the reference's literal step-halving rule.
                cpu = not args.no_cpu_baseline
                nb3 = min(3, len(bits))
                res["extra"] = {
                    "C2N5000": fixed_effects_line("C2N5000", dev, local, cpu=cpu, parity=not args.no_parity),
                    "C4": fixed_effects_line("C4", dev, local, cpu=cpu, parity=not args.no_parity),
                    "C4_literal": fixed_effects_line("C4", dev, local, cpu=False, parity=not args.no_parity, env={"SEERHIP_ROUTE": "firth_literal=1"}),
                    "C2": fixed_effects_line("C2", dev, local, steps=3, cpu=cpu, parity=not args.no_parity),
                    # a caller that holds ONE batch at a time (no lanes to fill): what a large call buys -- the `synchronous` entry of these two is
                    # one sh_glm_batch_dev of 2^20 rows per step on one stream, beside the same rows through the lanes
                    "C2N5000_calls_of_2e20_rows": fixed_effects_line("C2N5000", dev, local, steps=3, Vs=1 << 20, cpu=False, parity=not args.no_parity),
                    "C4_calls_of_2e20_rows": fixed_effects_line("C4", dev, local, steps=3, Vs=1 << 20, cpu=False, parity=not args.no_parity),
                    "C3_five_limbs": lmm_variant_line(U, S, y, C, h2, dev, local, bits[:nb3], 3, limbs=5),
                    "C3_all_refined_56bit": lmm_variant_line(U, S, y, C, h2, dev, local, bits[:nb3], 3, limbs=0, tol=1e-300),
                    "C3_host_pointers_pcie_inclusive": lmm_host_pointer_line(U, S, y, C, h2, dev, local, bits[0], 4),
                }
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline_lmm(U, S, y, C, h2, N)
        else:
            force = cfg == "C4"
            metric, dtype = glm_metric(cfg, N)
            res.update({"metric": metric, "dtype": dtype, "config": glm_workload(cfg, Vs, N, q), "lanes": lanes, "contexts": 1,
                        "roofline": with_clock(glm_roofline(cfg, q, Vs, kern_s, klaunch, rb), clocks)})
            res["config"]["workload"] += ", one engine context, batches through its %d lanes" % lanes
            res["roofline"]["batch_span_ms"] = kms / max(klaunch, 1)
            res["roofline"]["kernel_ms_is"] = "timed region / steps (the batches of different lanes overlap; batch_span_ms = one batch's own HIP-event span on its lane's stream)"
            if not args.no_parity:
                n, devs = parity_glm(y, W, nl, nf, force, last, out, fl, N)
                res["parity_checked"] = n; res["parity_max_rel_dev"] = devs
            if world == 1 and not args.no_extra:
                # the same context, one synchronous sh_glm_batch_dev per step on one stream (rounds 1-4's measurement), and its rows for the
                # last timed batch: the lanes' must be the same bytes
                try:
                    eng.set_timing(False)
                    o1 = torch.empty_like(out); f1 = torch.empty_like(fl)
                    eng.glm_batch_dev(last, o1, f1); torch.cuda.synchronize()
                    res["identical_to_synchronous_call"] = bool(torch.equal(o1.view(torch.int64), out.view(torch.int64))) and bool(torch.equal(f1, fl))
                    eng.set_timing(True)
                    ns = max(3, min(args.steps, 6))
                    t1 = time.perf_counter()
                    for i in range(ns):
                        eng.glm_batch_dev(bits[i % ndist], o1, f1)
                    torch.cuda.synchronize()
                    d1 = time.perf_counter() - t1
                    k1, n1 = eng.get_timing()
                    res["synchronous"] = {"value": Vs * ns / d1, "unit": "variants/s", "steps": ns, "ms_per_step": d1 / ns * 1e3,
                                          "roofline": glm_roofline(cfg, q, Vs, k1 / max(n1, 1) * 1e-3, n1, rb)}
                    del o1, f1
                except Exception as e:
                    res["synchronous"] = {"error": repr(e)}
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline_glm(y, W, nl, nf, N, force)
        emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
