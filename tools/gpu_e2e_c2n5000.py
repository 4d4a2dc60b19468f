"""C2N5000 as a user runs it: `python -m pyseer_amd --no-distances --covariates cov.tsv --use-covariates 2q .. 11q --load-packed kmers.seerpack`
(fixed effects, logistic with Firth for the routed rows, 10 quantitative covariates) at N = 5000 over the packed k-mers tools/gpu_e2e_c3.py wrote
(run it first in the same gpurun call: /tmp/e2e_c3 stays), default --lrt-pvalue 1: every tested row printed with its 10 covariate slopes.
Phenotype and covariates are bench.py's C2N5000 workload (synth_glm_inputs).  Reports what gpu_e2e_c3.py reports."""
import json, os, subprocess, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench

N = 5000; BLK = int(os.environ.get("BLOCK", 262144))
src = os.environ.get("E2E_DIR", "/tmp/e2e_c3"); d = "/tmp/e2e_c2"; os.makedirs(d, exist_ok=True)
assert os.path.exists(src + "/kmers.seerpack"), "run tools/gpu_e2e_c3.py first (same gpurun call)"
names = ["sample_%05d" % i for i in range(N)]
y, W, _, _ = bench.synth_glm_inputs(N, 10)
with open(d + "/pheno.tsv", "w") as f:
    f.write("samples\tbinary\n")
    for i in range(N):
        f.write("%s\t%d\n" % (names[i], int(y[i])))
with open(d + "/cov.tsv", "w") as f:
    f.write("samples\t" + "\t".join("c%d" % j for j in range(10)) + "\n")
    for i in range(N):
        f.write(names[i] + "\t" + "\t".join(repr(float(x)) for x in W[i]) + "\n")
V = None
env0 = dict(os.environ); env0["PYTHONPATH"] = ROOT; env0["SEERHIP_DEBUG"] = "cli"
res = {"n_samples": N, "block_size": BLK, "covariates": 10, "cache_GB": os.path.getsize(src + "/kmers.seerpack") / 1e9}
runs = [("overlapped", []), ("serial", ["--serial-sink"])]
if os.environ.get("E2E_EXTRA"):                              # e.g. "--gpus 0,0": a probe, not part of the record
    runs.append(("overlapped_extra", os.environ["E2E_EXTRA"].split()))
for name, extra in runs:
    out = d + "/out_%s.tsv" % name
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "pyseer_amd", "--kmers", src + "/kmers.txt", "--uncompressed", "--phenotypes", d + "/pheno.tsv", "--no-distances",
                        "--covariates", d + "/cov.tsv", "--use-covariates"] + ["%dq" % j for j in range(2, 12)] +
                       ["--load-packed", src + "/kmers.seerpack", "--block_size", str(BLK), "--no-dedup"] + extra,
                       env=env0, stdout=open(out, "w"), stderr=subprocess.PIPE)
    dt = time.time() - t0
    err = r.stderr.decode()
    tl = [l for l in err.splitlines() if l.startswith("[cli timing]")]
    loaded = [l for l in err.splitlines() if l.endswith("loaded variants")]
    V = int(loaded[-1].split()[0]) if loaded else 0
    print("%s: rc %d, %.1f s wall, %.3g rows/s end to end (start-up included), output %.2f GB" % (name, r.returncode, dt, V / dt, os.path.getsize(out) / 1e9))
    for l in tl[-3:]:
        print("   ", l)
    print("   ", err.strip().splitlines()[-4:])
    res[name] = {"rc": r.returncode, "wall_s": dt, "rows_per_s_wall": V / dt, "output_GB": os.path.getsize(out) / 1e9, "timing": tl[-3:]}
res["k_mers"] = V
def digest(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for c in iter(lambda: f.read(1 << 24), b""):
            h.update(c)
    return h.hexdigest()
res["outputs_identical"] = digest(d + "/out_overlapped.tsv") == digest(d + "/out_serial.tsv")
print("overlapped and serial outputs identical:", res["outputs_identical"])
print(open(d + "/out_overlapped.tsv").readline().rstrip()[:300])
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "r03"); os.makedirs(o, exist_ok=True)
json.dump(res, open(o + "/e2e_c2n5000.json", "w"), indent=1)
