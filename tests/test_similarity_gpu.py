"""Similarity (kinship) accumulation on the device vs the numpy restatement and the reference's own output
(tests/golden/cli/expected/similarity_*.tsv were written by the reference's `similarity` script, make_cli_golden.py)."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu
CLI = os.path.join(os.path.dirname(__file__), "golden", "cli")


def _engine(n):
    from pyseer_amd.engine import Engine
    return Engine(n)


@pytest.mark.parametrize("n,V,seed", [(50, 200, 1), (300, 5000, 2), (1000, 40000, 3), (777, 12345, 4), (129, 70001, 5)])
def test_similarity_matches_oracle(n, V, seed):
    from oracle import oracle
    from pyseer_amd.packing import pack_variants
    rng = np.random.default_rng(seed)
    af = rng.uniform(0.0, 1.0, V)
    Kv = (rng.random((V, n)) < af[:, None]).astype(np.uint8)
    Kv[0] = 0; Kv[1] = 1
    e = _engine(n)
    e.sim_begin()
    bits = pack_variants(Kv)
    cut = V // 3
    e.sim_accumulate(bits[:cut]); e.sim_accumulate(bits[cut:])          # accumulation over batches
    K = e.sim_finish()
    assert np.array_equal(K, oracle.similarity(Kv))
    # device-side allele-frequency filter
    e.set_af_filter(0.05, 0.9)
    e.sim_begin(); e.sim_accumulate(bits)
    assert np.array_equal(e.sim_finish(), oracle.similarity(Kv, 0.05, 0.9))


def test_similarity_begin_resets():
    from pyseer_amd.packing import pack_variants
    Kv = np.ones((10, 64), dtype=np.uint8)
    e = _engine(64)
    e.sim_begin(); e.sim_accumulate(pack_variants(Kv)); e.sim_begin(); e.sim_accumulate(pack_variants(Kv[:3]))
    assert np.array_equal(e.sim_finish(), np.full((64, 64), 3.0))


def _run_cli(argv, capsys):
    from pyseer_amd import similarity
    similarity.main(argv)
    return capsys.readouterr()


@pytest.mark.parametrize("name,args", [
    ("similarity_kmers", ["--kmers", os.path.join(CLI, "kmers.gz")]),
    ("similarity_kmers", ["--kmers", os.path.join(CLI, "kmers.gz"), "--python-reader"]),
    ("similarity_rtab", ["--pres", os.path.join(CLI, "kmers120.Rtab"), "--min-af", "0.1", "--max-af", "0.8"]),
])
def test_similarity_cli_matches_reference(name, args, capsys):
    out = _run_cli([os.path.join(CLI, "samples50.txt")] + args, capsys)
    exp = open(os.path.join(CLI, "expected", name + ".tsv")).read()
    assert out.out == exp
    exp_err = open(os.path.join(CLI, "expected", name + ".err")).read()
    assert out.err == exp_err


def test_similarity_missing_calls_give_nan_rows():
    """np.matmul over a G holding NaN (missing Rtab/VCF calls kept under --max-missing) poisons those samples' rows/columns."""
    from pyseer_amd.similarity import SimilarityAccumulator
    rng = np.random.default_rng(7)
    n, V = 70, 300
    Kv = (rng.random((V, n)) < 0.4).astype(float)
    Kv[5, 3] = np.nan; Kv[17, 66] = np.nan; Kv[17, 3] = np.nan
    acc = SimilarityAccumulator(n)
    for k in Kv:
        acc.add_dense(k)
    K = acc.finish()
    with np.errstate(invalid="ignore"):
        ref = np.matmul(Kv.T, Kv)
    assert np.array_equal(np.isnan(K), np.isnan(ref))
    assert np.array_equal(K[~np.isnan(ref)], ref[~np.isnan(ref)])
