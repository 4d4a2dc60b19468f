"""Stream contracts of the reference recorded on the k-mer fixture: iter_variants' 16-tuples and load_var_block's blocks
(pyseer/input.py:505-707).  Run: PYTHONPATH=_harness:_harness/stubs:/root/reference /opt/conda/bin/python3.9 -W ignore make_stream_golden.py"""
import shim  # noqa: F401
import gzip, json, os
import numpy as np
import pandas as pd
from pyseer.input import iter_variants, load_var_block, load_phenotypes

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(HERE, "cli")
p = load_phenotypes(os.path.join(CLI, "subset.pheno"), None)
all_strains = set(p.index)
m = np.zeros((len(p), 0)); cov = pd.DataFrame([])
out = {"iter_variants": [], "load_var_block": []}
with gzip.open(os.path.join(CLI, "kmers.gz"), "r") as fh:
    it = iter_variants(p, m, cov, "kmers", False, None, fh, all_strains, [], False, None, 0.05, 0.9, 0.05, 0.7, 0.3, "NULL", "FIRTH",
                       False, False)
    for n, t in enumerate(it):
        if n >= 60:
            break
        (name, pv, k, m_, c_, af, pattern, lin_eff, lin_cl, fp, lp, nf, ff, ks, nks, cont) = t
        out["iter_variants"].append(dict(name=name, p_is_none=pv is None, k=None if k is None else [int(x) for x in k], af=float(af),
                                         pattern=None if pattern is None else pattern.decode(), filter_pvalue=fp, lrt_pvalue=lp,
                                         null_fit=nf, firth_null=ff, kstrains=list(ks), nkstrains=list(nks), continuous=cont,
                                         lineage_effects=lin_eff))
with gzip.open(os.path.join(CLI, "kmers.gz"), "r") as fh:
    it = load_var_block("kmers", p, False, None, fh, all_strains, [], 0.05, 0.9, 0.05, False, 25)
    for n, (variants, mat, eof) in enumerate(it):
        if variants is None or n >= 3:
            out["load_var_block"].append(dict(terminal=variants is None, eof=bool(eof)))
            break
        out["load_var_block"].append(dict(
            eof=bool(eof), shape=list(mat.shape), colsum=[float(x) for x in mat.sum(axis=0)],
            variants=[dict(kmer=v[0].kmer, pattern=None if v[0].pattern is None else v[0].pattern.decode(), af=float(v[0].af),
                           prefilter=v[0].prefilter, filter=v[0].filter, notes=sorted(v[0].notes), n_k=int(np.nansum(v[2])))
                      for v in variants]))
json.dump(out, open(os.path.join(HERE, "stream_contracts.json"), "w"), indent=0)
print(len(out["iter_variants"]), "tuples;", len(out["load_var_block"]), "blocks")
