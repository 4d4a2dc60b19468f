"""End-to-end timing of the command line on a synthetic gzip k-mer file (N samples x V k-mers): where does a real run spend its
time?  Writes the inputs under /tmp, runs `python -m pyseer_amd --lmm`, prints wall-clock per stage (stderr of the CLI is kept)."""
import gzip, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(os.environ.get("N", 2000)); V = int(os.environ.get("V", 20000))
rng = np.random.default_rng(0)
names = ["sample_%05d" % i for i in range(N)]
d = "/tmp/e2e"; os.makedirs(d, exist_ok=True)
t0 = time.time()
lin = rng.integers(0, 20, N)
G = (rng.random((600, N)) < (0.1 + 0.6 * rng.random((600, 20)))[:, lin]).astype(np.float32)
K = G.T @ G
with open(d + "/sim.tsv", "w") as f:
    f.write("\t" + "\t".join(names) + "\n")
    for i in range(N):
        f.write(names[i] + "\t" + "\t".join("%g" % x for x in K[i]) + "\n")
y = (rng.random(N) < 0.2 + 0.5 * (lin < 8)).astype(int)
with open(d + "/pheno.tsv", "w") as f:
    f.write("samples\tbinary\n")
    for i in range(N):
        f.write("%s\t%d\n" % (names[i], y[i]))
NF = int(os.environ.get("FILES", 4))                        # the same lines also as NF files, for `--kmers part0.gz part1.gz ...`
tok = np.array([n + ":1" for n in names], dtype=object)
parts = [gzip.open(d + "/part%d.gz" % i, "wt", compresslevel=4) for i in range(NF)]
with gzip.open(d + "/kmers.gz", "wt", compresslevel=4) as f:
    for v in range(V):
        af = rng.uniform(0.02, 0.98)
        idx = np.nonzero(rng.random(N) < af)[0]
        line = "".join(rng.choice(list("ACGT"), 31)) + " | " + " ".join(tok[idx]) + "\n"
        f.write(line); parts[v * NF // V].write(line)
for fh in parts:
    fh.close()
print("inputs written in %.1f s (%.1f MB gz)" % (time.time() - t0, os.path.getsize(d + "/kmers.gz") / 1e6))
env = dict(os.environ); env["PYTHONPATH"] = ROOT
t0 = time.time()
r = subprocess.run([sys.executable, "-m", "pyseer_amd", "--kmers"] + [d + "/part%d.gz" % i for i in range(NF)] + ["--phenotypes", d + "/pheno.tsv", "--lmm",
                    "--similarity", d + "/sim.tsv", "--block_size", "65536"], env=env, stdout=open(d + "/out_parts.tsv", "w"), stderr=subprocess.PIPE)
dt = time.time() - t0
print("CLI %d files, block 65536: rc %d %.1f s total -> %.0f k-mers/s end to end" % (NF, r.returncode, dt, V / dt))
for extra in ([], ["--block_size", "65536", "--save-packed", d + "/kmers.pk"], ["--block_size", "65536", "--load-packed", d + "/kmers.pk"],
              ["--save-lmm", d + "/lmm.npz", "--block_size", "65536", "--load-packed", d + "/kmers.pk"]):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "pyseer_amd", "--kmers", d + "/kmers.gz", "--phenotypes", d + "/pheno.tsv", "--lmm",
                        "--similarity", d + "/sim.tsv"] + extra, env=env, stdout=open(d + "/out.tsv", "w"), stderr=subprocess.PIPE)
    dt = time.time() - t0
    print("CLI", extra, "rc", r.returncode, "%.1f s total -> %.0f k-mers/s end to end" % (dt, V / dt))
    print("   ", r.stderr.decode().strip().splitlines()[-4:])
    if extra == ["--block_size", "65536", "--save-packed", d + "/kmers.pk"]:
        same = open(d + "/out.tsv").read() == open(d + "/out_parts.tsv").read()
        print("    output of the %d-file run identical to the one-file run at the same block size: %s" % (NF, same))

# ---- steady state from a packed cache: BIGV synthetic variants written straight into the cache format (no text involved)
BIGV = int(os.environ.get("BIGV", 0))
if BIGV:
    sys.path.insert(0, ROOT)
    from pyseer_amd.input import PackedCacheWriter
    from pyseer_amd.packing import row_bytes_for
    t0 = time.time()
    w = PackedCacheWriter(d + "/big.pk", names)
    rb = row_bytes_for(N)
    blk = 1 << 17
    mask = np.zeros(rb, dtype=np.uint8); mask[:N // 8] = 255
    if N % 8: mask[N // 8] = (1 << (N % 8)) - 1
    for s in range(0, BIGV, blk):
        nv = min(blk, BIGV - s)
        bits = rng.integers(0, 256, (nv, rb), dtype=np.uint8) & mask
        counts = np.unpackbits(bits, axis=1).sum(axis=1).astype(np.int32)
        nm = ("".join(rng.choice(list("ACGT"), 31 * nv))).encode()
        off = np.arange(nv + 1, dtype=np.int64) * 31
        w.write_block(nm, off, counts, bits)
    w.close()
    print("cache with %d variants written in %.1f s (%.2f GB)" % (BIGV, time.time() - t0, os.path.getsize(d + "/big.pk") / 1e9))
    for extra in (["--block_size", "262144"], ["--block_size", "262144", "--no-dedup"]):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "pyseer_amd", "--kmers", d + "/kmers.gz", "--phenotypes", d + "/pheno.tsv", "--lmm",
                            "--similarity", d + "/sim.tsv", "--load-packed", d + "/big.pk"] + extra, env=env,
                           stdout=open(d + "/out_big.tsv", "w"), stderr=subprocess.PIPE)
        dt = time.time() - t0
        print("CLI from cache", extra, "rc", r.returncode, "%.1f s total -> %.0f k-mers/s end to end; output %.2f GB"
              % (dt, BIGV / dt, os.path.getsize(d + "/out_big.tsv") / 1e9))
        print("   ", r.stderr.decode().strip().splitlines()[-4:])
