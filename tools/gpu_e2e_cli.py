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
with gzip.open(d + "/kmers.gz", "wt", compresslevel=4) as f:
    for v in range(V):
        af = rng.uniform(0.02, 0.98)
        idx = np.nonzero(rng.random(N) < af)[0]
        f.write("".join(rng.choice(list("ACGT"), 31)) + " | " + " ".join(names[i] + ":1" for i in idx) + "\n")
print("inputs written in %.1f s (%.1f MB gz)" % (time.time() - t0, os.path.getsize(d + "/kmers.gz") / 1e6))
env = dict(os.environ); env["PYTHONPATH"] = ROOT
for extra in ([], ["--block_size", "65536"]):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "pyseer_amd", "--kmers", d + "/kmers.gz", "--phenotypes", d + "/pheno.tsv", "--lmm",
                        "--similarity", d + "/sim.tsv"] + extra, env=env, stdout=open(d + "/out.tsv", "w"), stderr=subprocess.PIPE)
    dt = time.time() - t0
    print("CLI", extra, "rc", r.returncode, "%.1f s total -> %.0f k-mers/s end to end" % (dt, V / dt))
    print("   ", r.stderr.decode().strip().splitlines()[-4:])
