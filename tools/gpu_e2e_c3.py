"""C3 as a user runs it: `python -m pyseer_amd --lmm --load-lmm cache.npz --load-packed kmers.seerpack` at N = 5000 over V packed k-mers
(default 10 M = a 6.4 GB cache), default --lrt-pvalue 1 (every tested row printed), to a TSV file.  The inputs are synthesised on the box:
the kinship decomposition of bench.py's C3 workload saved as the reference's --save-lmm cache, the k-mer rows generated on the GPU in the
bench's AF mix and written in the packed-cache format (no text involved: this measures the steady state a second run over a k-mer file sees).
Reports rows/s end to end and the split of a block's time into engine call (H2D + GPU + D2H) / sink (masking, formatting) / write, measured
with SEERHIP_DEBUG=cli; runs the default (overlapped) loop and --serial-sink, and compares their output bytes."""
import json, os, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pyseer_amd.input import PackedCacheWriter
from pyseer_amd.packing import row_bytes_for

N = 5000; V = int(os.environ.get("V", 10_000_000)); BLK = int(os.environ.get("BLOCK", 262144))
LRT = ["--lrt-pvalue", os.environ["E2E_LRT"]] if os.environ.get("E2E_LRT") else []
d = os.environ.get("E2E_DIR", "/tmp/e2e_c3"); os.makedirs(d, exist_ok=True)
dev = torch.device("cuda", 0)
names = ["sample_%05d" % i for i in range(N)]
t0 = time.time()
U, S, h2, C, y, lin = bench.synth_lmm_inputs(N, 1003, dev)
np.savez(d + "/lmm.npz", U, S, np.array([h2]))                 # pyseer --save-lmm layout: arr_0 = U, arr_1 = S, arr_2 = [h2] (pyseer/lmm.py:66-70, 116-118)
with open(d + "/pheno.tsv", "w") as f:
    f.write("samples\tbinary\n")
    for i in range(N):
        f.write("%s\t%d\n" % (names[i], int(y[i])))
rb = row_bytes_for(N)
w = PackedCacheWriter(d + "/kmers.seerpack", names)
rng = np.random.default_rng(0)
alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
for s in range(0, V, 1 << 18):
    nv = min(1 << 18, V - s)
    bits = bench.synth_bits(nv, N, rb, 7000 + s, dev).cpu().numpy()
    counts = np.unpackbits(bits, axis=1).sum(axis=1).astype(np.int32)
    nm = alphabet[rng.integers(0, 4, 31 * nv)].tobytes()
    w.write_block(nm, np.arange(nv + 1, dtype=np.int64) * 31, counts, bits)
w.close()
open(d + "/kmers.txt", "w").write("AAAA | sample_00000:1\n")    # the CLI wants a k-mer file argument; --load-packed supplies the rows
print("inputs: %d k-mers x %d samples, cache %.2f GB, written in %.1f s" % (V, N, os.path.getsize(d + "/kmers.seerpack") / 1e9, time.time() - t0))
del U
torch.cuda.empty_cache()
env0 = dict(os.environ); env0["PYTHONPATH"] = ROOT; env0["SEERHIP_DEBUG"] = "cli"
res = {"n_samples": N, "k_mers": V, "block_size": BLK, "cache_GB": os.path.getsize(d + "/kmers.seerpack") / 1e9}
runs = [("overlapped", [], {}), ("serial", ["--serial-sink"], {})]
if os.environ.get("E2E_GPUS"):            # e.g. "0,0": two contexts on one device = the multi-device job path (one pipelined stream per context over its
    runs.append(("gpus_" + os.environ["E2E_GPUS"].replace(",", "_"), ["--gpus", os.environ["E2E_GPUS"]], {}))      # own range of the cache) on a one-GPU box
for t in [x for x in os.environ.get("E2E_SINK_THREADS", "").split(",") if x]:   # extra overlapped runs with the sink's OpenMP team capped (a probe, not part of the record)
    runs.append(("overlapped_omp%s" % t, [], {"OMP_NUM_THREADS": t}))
for name, extra, more_env in runs:
    out = d + "/out_%s.tsv" % name
    env = dict(env0); env.update(more_env)
    t0 = time.time(); ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    r = subprocess.run([sys.executable, "-m", "pyseer_amd", "--kmers", d + "/kmers.txt", "--uncompressed", "--phenotypes", d + "/pheno.tsv", "--lmm",
                        "--load-lmm", d + "/lmm.npz", "--load-packed", d + "/kmers.seerpack", "--block_size", str(BLK), "--no-dedup"] + LRT + extra,
                       env=env, stdout=open(out, "w"), stderr=subprocess.PIPE)
    dt = time.time() - t0; ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    err = r.stderr.decode()
    tl = [l for l in err.splitlines() if l.startswith("[cli timing]")]
    print("%s: rc %d, %.1f s wall, %.3g rows/s end to end (start-up included), output %.2f GB" % (name, r.returncode, dt, V / dt, os.path.getsize(out) / 1e9))
    for l in tl[-9:]:
        print("   ", l)
    print("   ", err.strip().splitlines()[-4:])
    res[name] = {"rc": r.returncode, "wall_s": dt, "rows_per_s_wall": V / dt, "output_GB": os.path.getsize(out) / 1e9, "timing": tl[-9:],
                 "cpu_s_whole_process": cpu_s, "cpu_s_per_million_rows_whole_process": cpu_s / (V / 1e6)}
    print("    whole process: %.2f CPU-s (user + sys, all threads, start-up included) = %.4f per million rows" % (cpu_s, cpu_s / (V / 1e6)))
import hashlib
def digest(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for c in iter(lambda: f.read(1 << 24), b""):
            h.update(c)
    return h.hexdigest()
res["outputs_identical"] = digest(d + "/out_overlapped.tsv") == digest(d + "/out_serial.tsv")
print("overlapped and serial outputs identical:", res["outputs_identical"])
for name, extra, _ in runs[2:]:
    if name.startswith("gpus_"):
        res[name]["identical_to_single_engine"] = digest(d + "/out_%s.tsv" % name) == digest(d + "/out_overlapped.tsv")
        res[name]["rows_per_s_vs_single_engine"] = res[name]["rows_per_s_wall"] / res["overlapped"]["rows_per_s_wall"]
        print(name, "identical to the single-engine output:", res[name]["identical_to_single_engine"], " wall ratio %.3f" % res[name]["rows_per_s_vs_single_engine"])
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", os.environ.get("E2E_ROUND", "r05")); os.makedirs(o, exist_ok=True)
json.dump(res, open(o + "/e2e_c3.json", "w"), indent=1)
