"""The host budget of the `--gpus` job (VERDICT r04 "next" 1): C3 as a user runs it from a packed cache --
`python -m pyseer_amd --lmm --load-lmm cache.npz --load-packed kmers.seerpack [--lrt-pvalue 1e-3] [--gpus 0,0,0,0,0,0,0,0]` at N = 5000 over V
packed k-mers -- with one context and with EIGHT contexts on the one device of the box (the only stand-in for an 8-GPU node), through the
round-4 sink (SEERHIP_ROUTE job=0) and through the job stream, output compared byte for byte.  For every run: wall time, rows/s, and the CPU
seconds the block loop cost (process getrusage over the loop, the library's per-stage thread CPU, the loop threads' CPU: the "[cli budget]"
line of pyseer_amd/__main__.py).  Writes gpurun_out/r05/host_budget.json:
    cpu_s_per_million_rows (by stage)  and  cpus_needed_at_8x33M = cpu_s_per_million_rows x 8 x 33 (million rows/s) against the box's quota."""
import hashlib, json, os, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pyseer_amd.input import PackedCacheWriter
from pyseer_amd.packing import row_bytes_for

N = 5000; V = int(os.environ.get("V", 10_000_000)); BLK = int(os.environ.get("BLOCK", 262144))       # BLOCK=0: the command line's own default
STORED = int(os.environ.get("STORED", 1 << 18))        # rows per stored block of the cache (round 5's --save-packed default wrote 2^16)
FIXED = os.environ.get("E2E_MODEL", "lmm") == "fixed"        # the fixed-effects job (logistic, 10 covariates: bench.py's C2N5000 workload) instead of --lmm
d = os.environ.get("E2E_DIR", "/tmp/e2e_job"); os.makedirs(d, exist_ok=True)
GPUS8 = os.environ.get("E2E_GPUS", "0,0,0,0,0,0,0,0")
dev = torch.device("cuda", 0)
names = ["sample_%05d" % i for i in range(N)]
t0 = time.time()
U, S, h2, C, y, lin = bench.synth_lmm_inputs(N, 1003, dev)
np.savez(d + "/lmm.npz", U, S, np.array([h2]))                 # pyseer --save-lmm layout (pyseer/lmm.py:66-70, 116-118)
if FIXED:
    y, Wc, _, _ = bench.synth_glm_inputs(N, 10)
    with open(d + "/cov.tsv", "w") as f:
        f.write("samples\t" + "\t".join("c%d" % j for j in range(10)) + "\n")
        for i in range(N):
            f.write(names[i] + "\t" + "\t".join(repr(float(x)) for x in Wc[i]) + "\n")
# (round 6: --lineage / --output-patterns / --print-samples inside the job stream -- E2E_EXTRA="--lineage --lineage-clusters {d}/clusters.txt
# --lineage-file {d}/lin.txt --output-patterns {d}/patterns.txt"; clusters = the lineages the synthetic kinship was built from, 8 of them)
with open(d + "/clusters.txt", "w") as f:
    for i in range(N):
        f.write("%s\tcl%d\n" % (names[i], int(lin[i]) % 8))
EXTRA = [x.replace("{d}", d) for x in os.environ.get("E2E_EXTRA", "").split()]
with open(d + "/pheno.tsv", "w") as f:
    f.write("samples\tbinary\n")
    for i in range(N):
        f.write("%s\t%d\n" % (names[i], int(y[i])))
rb = row_bytes_for(N)
w = PackedCacheWriter(d + "/kmers.seerpack", names)
rng = np.random.default_rng(0)
alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
for s in range(0, V, STORED):
    nv = min(STORED, V - s)
    bits = bench.synth_bits(nv, N, rb, 7000 + s, dev).cpu().numpy()
    counts = np.unpackbits(bits, axis=1).sum(axis=1).astype(np.int32)
    nm = alphabet[rng.integers(0, 4, 31 * nv)].tobytes()
    w.write_block(nm, np.arange(nv + 1, dtype=np.int64) * 31, counts, bits)
w.close()
open(d + "/kmers.txt", "w").write("AAAA | sample_00000:1\n")
print("inputs: %d k-mers x %d samples, cache %.2f GB, written in %.1f s" % (V, N, os.path.getsize(d + "/kmers.seerpack") / 1e9, time.time() - t0), flush=True)
del U
torch.cuda.empty_cache()
quota = None
try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()
    quota = None if q == "max" else float(q) / float(per)
except Exception:
    pass
env0 = dict(os.environ); env0["PYTHONPATH"] = ROOT; env0["SEERHIP_DEBUG"] = "cli"
res = {"n_samples": N, "k_mers": V, "block_size": BLK, "stored_block_rows": STORED, "cache_GB": os.path.getsize(d + "/kmers.seerpack") / 1e9, "cpu_quota": quota, "nproc": os.cpu_count(), "runs": {}}


def digest(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for c in iter(lambda: f.read(1 << 24), b""):
            h.update(c)
    return h.hexdigest()


def run(name, extra, env_more):
    out = d + "/out_%s.tsv" % name
    env = dict(env0); env.update(env_more)
    t0 = time.time(); ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    model = (["--no-distances", "--covariates", d + "/cov.tsv", "--use-covariates"] + ["%dq" % j for j in range(2, 12)]) if FIXED else ["--lmm", "--load-lmm", d + "/lmm.npz"]
    r = subprocess.run([sys.executable, "-m", "pyseer_amd", "--kmers", d + "/kmers.txt", "--uncompressed", "--phenotypes", d + "/pheno.tsv"] + model +
                       ["--load-packed", d + "/kmers.seerpack"] + (["--block_size", str(BLK)] if BLK else []) + ["--no-dedup"] + EXTRA + extra,
                       env=env, stdout=open(out, "w"), stderr=subprocess.PIPE)
    dt = time.time() - t0; ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    err = r.stderr.decode()
    bl = [l for l in err.splitlines() if l.startswith("[cli budget] ")]
    budget = json.loads(bl[-1][len("[cli budget] "):]) if bl else None
    tl = [l for l in err.splitlines() if l.startswith("[cli timing]")]
    e = {"rc": r.returncode, "wall_s": dt, "rows_per_s_wall": V / dt, "output_bytes": os.path.getsize(out), "md5": digest(out),
         "process_cpu_s_incl_startup": (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime), "budget": budget, "timing": tl[-10:],
         "stderr_tail": err.strip().splitlines()[-4:]}
    if budget and budget["rows"]:
        m = budget["rows"] / 1e6
        e["loop_rows_per_s"] = budget["rows"] / budget["wall_s"]
        e["cpu_s_per_million_rows"] = {"process_total": budget["process_cpu_s"] / m, "process_user": budget["process_user_s"] / m, "process_sys": budget["process_sys_s"] / m}
        for k, v in budget["library_stage_cpu_s"].items():
            if v:
                e["cpu_s_per_million_rows"]["library_" + k] = v / m
        e["cpu_s_per_million_rows"]["loop_threads"] = sum(budget["loop_thread_cpu_s"].values()) / m
        e["cpus_needed_at_8x33M_rows_per_s"] = e["cpu_s_per_million_rows"]["process_total"] * 8 * 33.0
    res["runs"][name] = e
    print("%s: rc %d, %.2f s wall; loop %.3g rows/s; block loop %.3f CPU-s = %.4f CPU-s per million rows -> %.1f CPUs at 8 x 33 M rows/s (quota %s); output %d bytes" % (
        name, r.returncode, dt, e.get("loop_rows_per_s", float("nan")), budget["process_cpu_s"] if budget else float("nan"),
        e.get("cpu_s_per_million_rows", {}).get("process_total", float("nan")), e.get("cpus_needed_at_8x33M_rows_per_s", float("nan")), quota, e["output_bytes"]), flush=True)
    for l in tl[-10:]:
        print("    " + l[:400])
    if r.returncode:
        print(err[-3000:])
    return e


def run_procs(name, extra, env_more, nproc=8):
    """The same job as `nproc` PROCESSES side by side, each over its own range of the cache (--packed-part i/n) on the one device: what one
    process per GPU of an 8-GPU node costs the host (separate address spaces: no shared lock under page faults and hipHostRegister)."""
    env = dict(env0); env.update(env_more)
    model = (["--no-distances", "--covariates", d + "/cov.tsv", "--use-covariates"] + ["%dq" % j for j in range(2, 12)]) if FIXED else ["--lmm", "--load-lmm", d + "/lmm.npz"]
    t0 = time.time()
    procs = []
    for i in range(nproc):
        out = d + "/out_%s_%d.tsv" % (name, i)
        procs.append((subprocess.Popen([sys.executable, "-m", "pyseer_amd", "--kmers", d + "/kmers.txt", "--uncompressed", "--phenotypes", d + "/pheno.tsv"] + model +
                                       ["--load-packed", d + "/kmers.seerpack"] + (["--block_size", str(BLK)] if BLK else []) + ["--no-dedup", "--packed-part", "%d/%d" % (i, nproc)] + EXTRA + extra,
                                       env=env, stdout=open(out, "w"), stderr=subprocess.PIPE), out))
    errs = [p_.communicate()[1].decode() for p_, _ in procs]
    dt = time.time() - t0
    h = hashlib.md5(); nbytes = 0
    for i, (p_, out) in enumerate(procs):
        with open(out, "rb") as f:
            if i:
                f.readline()                                  # (every process prints the header)
            for c in iter(lambda: f.read(1 << 24), b""):
                h.update(c); nbytes += len(c)
    budgets = [json.loads([l for l in e_.splitlines() if l.startswith("[cli budget] ")][-1][len("[cli budget] "):]) for e_ in errs]
    rows = sum(b["rows"] for b in budgets); cpu = sum(b["process_cpu_s"] for b in budgets); wall = max(b["wall_s"] for b in budgets)
    stages = {}
    for b in budgets:
        for k, v in b["library_stage_cpu_s"].items():
            stages[k] = stages.get(k, 0.0) + v
    m = rows / 1e6
    e = {"rc": max(p_.returncode for p_, _ in procs), "processes": nproc, "wall_s": dt, "output_bytes": nbytes, "md5": h.hexdigest(), "loop_rows_per_s": rows / wall,
         "cpu_s_per_million_rows": dict({"process_total": cpu / m, "process_user": sum(b["process_user_s"] for b in budgets) / m,
                                         "process_sys": sum(b["process_sys_s"] for b in budgets) / m}, **{"library_" + k: v / m for k, v in stages.items() if v}),
         "cpus_needed_at_8x33M_rows_per_s": cpu / m * 8 * 33.0, "threads_by_name": [b.get("threads_by_name") for b in budgets][:2]}
    res["runs"][name] = e
    print("%s: rc %d, %.2f s wall; loops %.3g rows/s together; block loops %.3f CPU-s = %.4f CPU-s per million rows -> %.1f CPUs at 8 x 33 M rows/s (quota %s); output %d bytes" % (
        name, e["rc"], dt, e["loop_rows_per_s"], cpu, cpu / m, e["cpus_needed_at_8x33M_rows_per_s"], quota, nbytes), flush=True)
    if e["rc"]:
        print(errs[0][-2000:])
    return e


WHICH = os.environ.get("E2E_RUNS", "r04sink_1ctx,job_1ctx,r04sink_8ctx,job_8ctx").split(",")      # + job_8ctx_spin: the runtime's default (spinning) waits
for lrt, tag in [x.split(":") for x in os.environ.get("E2E_LRT", "1e-3:lrt1e-3,1:lrt1").split(",")]:
    L = ["--lrt-pvalue", lrt]
    md5 = set()
    for w_ in WHICH:
        extra = L + (["--gpus", GPUS8] if "8ctx" in w_ else [])
        rt = []
        if w_.startswith("r04sink"):
            rt.append("job=0")
        if w_.endswith("_spin"):
            rt.append("wait=spin")
        if w_.endswith("_staged"):
            rt.append("dma=0")
        if "_threads" in w_:                               # the devices' streams as threads of one process (round 5) instead of one process each
            rt.append("procs=0")
        if "_py" in w_:                                    # round 5's loop: the job stream driven block by block from Python
            rt.append("job=py")
        if "_lanes1" in w_:
            rt.append("lanes=1")
        env_more = {"SEERHIP_ROUTE": ",".join(rt)} if rt else {}
        if "8proc" in w_:
            md5.add(run_procs(w_ + "_" + tag, L, env_more)["md5"])
            continue
        md5.add(run(w_ + "_" + tag, extra, env_more)["md5"])
    res["identical_" + tag] = len(md5) == 1
    print("outputs identical (%s): %s" % (tag, res["identical_" + tag]), flush=True)
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", os.environ.get("E2E_ROUND", "r06")); os.makedirs(o, exist_ok=True)
json.dump(res, open(o + "/" + os.environ.get("E2E_OUT", "host_budget.json"), "w"), indent=1)
