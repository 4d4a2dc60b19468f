"""Host-side boundary contracts against vectors recorded from the reference (tests/golden/make_golden.py run_boundary,
tests/golden/make_stream_golden.py): format_output rows, hash_pattern digests, the 16-tuples of iter_variants and the blocks of
load_var_block (pyseer/utils.py:39-105, pyseer/input.py:505-723)."""
import gzip
import json
import os
import sys

import numpy as np
import pytest
import pandas as pd

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
G = os.path.join(os.path.dirname(__file__), "golden")
CLI = os.path.join(G, "cli")


def _num(x):
    return {"nan": np.nan, "inf": np.inf, "-inf": -np.inf}.get(x, x) if isinstance(x, str) else x


def test_format_output_rows():
    from pyseer_amd.classes import Seer, LMM
    from pyseer_amd.utils import format_output
    cases = json.load(open(os.path.join(G, "boundary.json")))["format_output"]
    assert len(cases) >= 10
    for c in cases:
        d = dict(c["item"])
        d["pattern"] = d["pattern"].encode()
        d["notes"] = set(d["notes"])
        for k in ("prep", "pvalue", "kbeta", "bse", "intercept", "frac_h2", "af"):
            if k in d:
                d[k] = _num(d[k])
        if "betas" in d:
            d["betas"] = np.array([np.nan if x is None else x for x in d["betas"]], dtype=float)
        item = (Seer if c["model"] == "seer" else LMM)(**d)
        got = format_output(item, c["lineage_dict"], c["model"], c["print_samples"])
        want = c["expected"]
        # the reference joins the notes set in hash order; rows are equal up to the order of the notes
        assert got.split("\t")[:-1] == want.split("\t")[:-1]
        assert sorted(got.split("\t")[-1].split(",")) == sorted(want.split("\t")[-1].split(","))


def test_hash_pattern_digests():
    from pyseer_amd.input import hash_pattern
    for h in json.load(open(os.path.join(G, "boundary.json")))["hash_pattern"]:
        k = np.array(h["k"], dtype=h["dtype"])
        assert hash_pattern(k).decode() == h["digest"]


def test_note_order_is_the_flag_bit_order():
    from pyseer_amd.classes import NOTE_ORDER
    assert list(NOTE_ORDER) == json.load(open(os.path.join(G, "boundary.json")))["note_order"]


def _pheno():
    from pyseer_amd.input import load_phenotypes
    return load_phenotypes(os.path.join(CLI, "subset.pheno"), None)


def test_iter_variants_tuples():
    from pyseer_amd.input import iter_variants
    want = json.load(open(os.path.join(G, "stream_contracts.json")))["iter_variants"]
    p = _pheno()
    m = np.zeros((len(p), 0)); cov = pd.DataFrame([])
    with gzip.open(os.path.join(CLI, "kmers.gz"), "r") as fh:
        it = iter_variants(p, m, cov, "kmers", False, None, fh, set(p.index), [], False, None, 0.05, 0.9, 0.05, 0.7, 0.3,
                           "NULL", "FIRTH", False, False)
        for w, t in zip(want, it):
            assert len(t) == 16
            (name, pv, k, m_, c_, af, pattern, lin_eff, lin_cl, fp, lp, nf, ff, ks, nks, cont) = t
            assert name == w["name"] and (pv is None) == w["p_is_none"] and af == w["af"]
            assert (None if k is None else [int(x) for x in k]) == w["k"]
            assert (None if pattern is None else pattern.decode()) == w["pattern"]
            assert (fp, lp, nf, ff, cont, lin_eff) == (w["filter_pvalue"], w["lrt_pvalue"], w["null_fit"], w["firth_null"],
                                                      w["continuous"], w["lineage_effects"])
            assert list(ks) == w["kstrains"] and list(nks) == w["nkstrains"]
            assert m_ is m and c_.shape == cov.values.shape           # m is passed through, c = cov.values (input.py:601)


def test_load_var_block_blocks():
    from pyseer_amd.input import load_var_block
    want = json.load(open(os.path.join(G, "stream_contracts.json")))["load_var_block"]
    p = _pheno()
    with gzip.open(os.path.join(CLI, "kmers.gz"), "r") as fh:
        it = load_var_block("kmers", p, False, None, fh, set(p.index), [], 0.05, 0.9, 0.05, False, 25)
        for w, (variants, mat, eof) in zip(want, it):
            if w.get("terminal"):
                assert variants is None and eof
                break
            if "variants" not in w:
                break
            assert bool(eof) == w["eof"] and list(mat.shape) == w["shape"]
            assert [float(x) for x in mat.sum(axis=0)] == w["colsum"]
            for (lm, pv, k), wv in zip(variants, w["variants"]):
                assert lm.kmer == wv["kmer"] and lm.af == wv["af"]
                assert (None if lm.pattern is None else lm.pattern.decode()) == wv["pattern"]
                assert (lm.prefilter, lm.filter, sorted(lm.notes)) == (wv["prefilter"], wv["filter"], wv["notes"])
                assert int(np.nansum(k)) == wv["n_k"]


def test_cmdscale_matches_the_references_own_vectors():
    """pyseer/cmdscale.py: the reference pins classical MDS on tests/distances_smaller.tsv.gz against two data files it ships
    (tests/cmdscale_test.py); the same files, copied as fixtures, pin the CLI's cmdscale here (axes are defined up to sign)."""
    from pyseer_amd.input import cmdscale
    F = os.path.join(G, "ref_fixtures")
    D = pd.read_csv(os.path.join(F, "distances_smaller.tsv.gz"), index_col=0, sep="\t")
    Y0 = np.loadtxt(os.path.join(F, "cmdscale.Y.txt.gz"))[:, :10]
    e0 = np.loadtxt(os.path.join(F, "cmdscale.e.txt.gz"))[:10]
    Y, e = cmdscale(D.values if hasattr(D, "values") else D)
    assert np.abs(np.abs(Y0) - np.abs(Y[:, :10])).max() < 1e-10
    assert np.abs(e0 - e[:10]).max() < 1e-10


def test_placeholder_sequence_and_torch_reachability_rule():
    """input._Repeat stands for a block's per-row placeholder lists (index, slice, iteration, len); __main__._torch_reachable decides from the raw
    arguments whether the command line can reach torch (only the kinship decomposition on the GPU does), prefixes as argparse accepts them."""
    import importlib
    from pyseer_amd.input import _Repeat
    r = _Repeat(b'', 5)
    assert len(r) == 5 and r[0] == b'' and r[-1] == b'' and r[np.int64(3)] == b'' and list(r) == [b''] * 5 and r[1:4] == [b''] * 3
    with pytest.raises(IndexError):
        r[5]
    assert [a for a, _ in zip(r, range(9))] == [b''] * 5
    m = importlib.import_module("pyseer_amd.__main__")
    reach = m._torch_reachable
    assert reach(["prog", "--lmm", "--similarity", "k.tsv"])
    assert not reach(["prog", "--lmm", "--load-lmm", "c.npz"]) and not reach(["prog", "--lmm", "--load-l", "c.npz"])
    assert not reach(["prog", "--lmm", "--similarity", "k.tsv", "--cpu-eigh"])
    assert not reach(["prog", "--kmers", "k.gz", "--no-distances"])
