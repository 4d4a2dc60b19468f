#!/opt/conda/bin/python3.9
"""Golden stdout/stderr of the REFERENCE command line for the C1 family (BASELINE.json configs[0]) -- build container only.

    cd /root/repo/tests/golden && PYTHONPATH=_harness:_harness/stubs:/root/reference PYTHONDONTWRITEBYTECODE=1 \
      OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 /opt/conda/bin/python3.9 -W ignore make_cli_golden.py

Inputs are the reference's own test data files, reduced to what the runs actually read (the reference subsets the
distance / similarity matrices to the phenotyped samples before using them, input.py:98-99, lmm.py:88-91), so the small
copies under cli/ give bit-identical runs.  Outputs: cli/expected/<name>.log|.err plus cli/cases.json.
"""
import gzip
import json
import os
import shutil
import subprocess
import sys

import pandas as pd

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(HERE, "cli")
EXP = os.path.join(CLI, "expected")
os.makedirs(EXP, exist_ok=True)

for f in ("kmers.gz", "subset.pheno", "covariates.txt"):
    shutil.copy(os.path.join(REF, f), os.path.join(CLI, f))
# An Rtab input with signal and with missing calls: the first 120 k-mers of kmers.gz re-expressed as a presence/absence
# table over the phenotyped samples ('.' = missing call, sprinkled deterministically).
_ph = pd.read_csv(os.path.join(REF, "subset.pheno"), index_col=0, sep="\t"); _ph.index = _ph.index.astype(str)
_samples = list(_ph.index)
with gzip.open(os.path.join(REF, "kmers.gz"), "rt") as fh, open(os.path.join(CLI, "kmers120.Rtab"), "w") as out:
    out.write("Gene\t" + "\t".join(_samples) + "\n")
    for i, line in enumerate(fh):
        if i >= 120:
            break
        name = line.split()[0]
        present = set(x.split(":")[0] for x in line.rstrip().split("|")[1].split())
        calls = ["1" if s_ in present else "0" for s_ in _samples]
        mid = 8 <= calls.count("1") <= len(calls) - 8     # (the reference crashes in chi2_contingency on degenerate tables)
        if mid and i % 3 == 1:
            calls[(7 * i) % len(calls)] = "."             # one missing call (2 % <= max-missing)
        if mid and i % 7 == 5:
            for j in range(5):
                calls[(3 * i + 11 * j) % len(calls)] = "."  # 10 % missing -> filtered
        out.write(name + "\t" + "\t".join(calls) + "\n")
p = pd.read_csv(os.path.join(REF, "subset.pheno"), index_col=0, sep="\t")
p.index = p.index.astype(str)
d = pd.read_csv(os.path.join(REF, "distances_not_square.tsv.gz"), index_col=0, sep="\t"); d.index = d.index.astype(str)
keep = p.index.intersection(d.index).intersection(d.columns)
d.loc[keep, keep].to_csv(os.path.join(CLI, "distances50.tsv"), sep="\t")
s = pd.read_csv(os.path.join(REF, "similarity.tsv.gz"), index_col=0, sep="\t"); s.index = s.index.astype(str)
keep = p.index.intersection(s.index)
s.loc[keep, keep].to_csv(os.path.join(CLI, "similarity50.tsv"), sep="\t")

# lineage clusters for the phenotyped samples: 4 groups assigned round-robin along the sample order of the distance matrix
with open(os.path.join(CLI, "clusters50.txt"), "w") as out:
    for i, name in enumerate(sorted(_samples)):
        out.write("%s\tBAPS_%d\n" % (name, 1 + (i * 7 % 11) % 4))
K, P, D, S, C, R = "kmers.gz", "subset.pheno", "distances50.tsv", "similarity50.tsv", "covariates.txt", "kmers120.Rtab"
CASES = {
    "fixed": ["--kmers", K, "--phenotypes", P, "--distances", D],
    "fixed_all": ["--kmers", K, "--phenotypes", P, "--distances", D, "--print-filtered"],
    "fixed_filters": ["--kmers", K, "--phenotypes", P, "--distances", D, "--filter-pvalue", "1E-2", "--lrt-pvalue", "1E-3"],
    "fixed_dim3_samples": ["--kmers", K, "--phenotypes", P, "--distances", D, "--max-dimensions", "3", "--print-samples"],
    "fixed_cov": ["--kmers", K, "--phenotypes", P, "--distances", D, "--max-dimensions", "3", "--covariates", C, "--use-covariates", "2q", "3"],
    "fixed_nodist": ["--kmers", K, "--phenotypes", P, "--no-distances", "--print-filtered"],
    "fixed_nodist_cov": ["--kmers", K, "--phenotypes", P, "--no-distances", "--use-covariates", "3", "--covariates", C],
    "cont": ["--kmers", K, "--phenotypes", P, "--distances", D, "--max-dimensions", "3", "--phenotype-column", "continuous", "--print-filtered"],
    "cont_forced": ["--kmers", K, "--phenotypes", P, "--distances", D, "--max-dimensions", "3", "--continuous"],
    "rtab": ["--pres", R, "--phenotypes", P, "--distances", D, "--max-dimensions", "3", "--print-filtered"],
    "fixed_lineage": ["--kmers", K, "--phenotypes", P, "--distances", D, "--max-dimensions", "3", "--lineage", "--lineage-file", "lineage_fixed.txt"],
    "fixed_lineage_clusters": ["--kmers", K, "--phenotypes", P, "--distances", D, "--max-dimensions", "3", "--lineage", "--lineage-clusters", "clusters50.txt", "--lineage-file", "lineage_clusters.txt"],
    "lmm_lineage": ["--kmers", K, "--phenotypes", P, "--similarity", S, "--lmm", "--lineage", "--distances", D, "--max-dimensions", "3", "--lineage-file", "lineage_lmm.txt"],
    "lmm": ["--kmers", K, "--phenotypes", P, "--similarity", S, "--lmm"],
    "lmm_all": ["--kmers", K, "--phenotypes", P, "--similarity", S, "--lmm", "--print-filtered", "--block_size", "50"],
    "lmm_cov": ["--kmers", K, "--phenotypes", P, "--similarity", S, "--lmm", "--covariates", C, "--use-covariates", "2q", "3"],
    "lmm_filters": ["--kmers", K, "--phenotypes", P, "--similarity", S, "--lmm", "--filter-pvalue", "0.2", "--lrt-pvalue", "0.5"],
    "lmm_rtab": ["--pres", R, "--phenotypes", P, "--similarity", S, "--lmm", "--print-filtered"],
    "lmm_cont": ["--kmers", K, "--phenotypes", P, "--similarity", S, "--lmm", "--phenotype-column", "continuous"],
}
env = dict(os.environ)
env["PYTHONPATH"] = os.pathsep.join([os.path.join(HERE, "_harness"), os.path.join(HERE, "_harness", "stubs"), "/root/reference"])
env.update(PYTHONDONTWRITEBYTECODE="1", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
for name, args in CASES.items():
    r = subprocess.run([sys.executable, "-W", "ignore", os.path.join(HERE, "_harness", "run_pyseer.py")] + args, cwd=CLI, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (name, r.stderr.decode()[-500:])
    open(os.path.join(EXP, name + ".log"), "wb").write(r.stdout)
    open(os.path.join(EXP, name + ".err"), "wb").write(r.stderr)
    print(name, len(r.stdout.splitlines()), "rows;", r.stderr.decode().strip().splitlines()[-4:])
    if "--lineage-file" in args:
        lf = args[args.index("--lineage-file") + 1]
        shutil.move(os.path.join(CLI, lf), os.path.join(EXP, lf))
json.dump(CASES, open(os.path.join(CLI, "cases.json"), "w"), indent=1)

# ---- the reference's `similarity` script (pyseer/similarity.py) on the same fixtures
shutil.copy(os.path.join(REF, "samples.txt"), os.path.join(CLI, "samples50.txt"))
SIM_CASES = {
    "similarity_kmers": ["samples50.txt", "--kmers", K],
    "similarity_rtab": ["samples50.txt", "--pres", R, "--min-af", "0.1", "--max-af", "0.8"],
}
for name, args in SIM_CASES.items():
    r = subprocess.run([sys.executable, "-W", "ignore", os.path.join(HERE, "_harness", "run_similarity.py")] + args, cwd=CLI, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (name, r.stderr.decode()[-500:])
    open(os.path.join(EXP, name + ".tsv"), "wb").write(r.stdout)
    open(os.path.join(EXP, name + ".err"), "wb").write(r.stderr)
