"""Native k-mer reader (pyseer_amd/csrc/reader.cpp) against the Python restatement of pyseer/input.py:read_variant, on the
reference's own kmers.gz and on synthetic gz / plain files with the format's corner cases."""
import gzip
import os

import numpy as np
import pandas as pd

from pyseer_amd.input import (NativeKmerReader, iter_packed_blocks, iter_packed_blocks_native, open_variant_file,
                              read_variant, hash_pattern)

CLI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli")


def _pheno():
    p = pd.read_csv(os.path.join(CLI, "subset.pheno"), index_col=0, sep="\t")["binary"]
    p.index = p.index.astype(str)
    return p


def test_native_equals_python_on_reference_kmers():
    p = _pheno()
    nat = list(iter_packed_blocks_native(p, os.path.join(CLI, "kmers.gz"), 0.01, 0.99, 37, want_patterns=True, want_samples=True))
    fh, _ = open_variant_file("kmers", os.path.join(CLI, "kmers.gz"))
    py = list(iter_packed_blocks(p, "kmers", fh, set(p.index), [], 0.01, 0.99, 0.05, False, 37))
    assert len(nat) == len(py) and sum(len(b.names) for b in nat) == 200
    for a, b in zip(nat, py):
        assert a.names == b.names and a.status == b.status and a.row_of == b.row_of
        assert np.array_equal(a.bits, b.bits) and np.allclose(a.afs, b.afs)
        assert a.patterns == b.patterns and a.kstrains == b.kstrains and a.nkstrains == b.nkstrains
        assert np.array_equal(a.last_k, np.asarray(b.last_k))


def test_corner_cases(tmp_path):
    samples = ["s%d" % i for i in range(70)] + ["dup"]
    lines = [
        "AAAA | s0:1 s1:3 s69:2",
        "CCCC\t|  s5:1   s5:7 unknown:1 dup:1 ",             # repeated sample, unknown sample, extra blanks
        "GGGG | s1:1 | s2:1",                                 # only the segment between the first two bars counts
        "TTTT |",                                             # no carriers
        "ACAC | " + " ".join("s%d:1" % i for i in range(70)),
    ]
    p = pd.Series(np.arange(len(samples), dtype=float), index=samples)
    for name, opener in (("k.txt", open), ("k.gz", gzip.open)):
        path = str(tmp_path / name)
        with opener(path, "wt") as fh:
            fh.write("\n".join(lines))                        # no trailing newline on purpose
        got = [(n, b.copy(), c.copy()) for n, b, c in NativeKmerReader(path, samples, 2)]
        names = sum((g[0] for g in got), [])
        bits = np.concatenate([g[1] for g in got]); counts = np.concatenate([g[2] for g in got])
        assert names == ["AAAA", "CCCC", "GGGG", "TTTT", "ACAC"]
        assert counts.tolist() == [3, 2, 1, 0, 70]
        # the Python restatement of the reference parser gives the same presence vectors
        fh = open(path) if name.endswith(".txt") else gzip.open(path, "r")
        for v in range(5):
            eof, k, var_name, ks, nks, af, missing = read_variant(fh, p, "kmers", False, None, name.endswith(".txt"),
                                                                  set(samples), [])
            dense = np.unpackbits(bits[v], bitorder="little")[:len(samples)]
            assert var_name == names[v] and np.array_equal(dense, k) and abs(af - counts[v] / len(samples)) < 1e-15
            assert hash_pattern(dense.astype(np.int64)) == hash_pattern(np.asarray(k))


def test_packed_cache_round_trip(tmp_path):
    """--save-packed / --load-packed: the cache reproduces the native reader's blocks (names, AFs, kept rows, order), also when
    stored blocks are merged into larger ones, and refuses another sample list."""
    import pandas as pd
    import pytest
    from pyseer_amd.input import iter_packed_blocks_native, iter_packed_blocks_cached, PackedCacheWriter
    cli = os.path.join(os.path.dirname(__file__), "golden", "cli")
    p = pd.read_csv(os.path.join(cli, "subset.pheno"), index_col=0, sep="\t")
    p.index = p.index.astype(str)
    p = p[p.columns[-1]]
    path = str(tmp_path / "k.pk")
    w = PackedCacheWriter(path, [str(x) for x in p.index])
    direct = list(iter_packed_blocks_native(p, os.path.join(cli, "kmers.gz"), 0.01, 0.99, 37, save_to=w))
    w.close()
    for bs in (37, 64, 100000):
        cached = list(iter_packed_blocks_cached(p, path, 0.01, 0.99, bs))
        names_d = [n for b in direct for n in b.names]
        names_c = [n for b in cached for n in b.names]
        assert names_d == names_c
        assert [a for b in direct for a in b.afs] == [a for b in cached for a in b.afs]
        assert [s for b in direct for s in b.status] == [s for b in cached for s in b.status]
        assert np.array_equal(np.concatenate([b.bits for b in direct]), np.concatenate([b.bits for b in cached]))
        for b in cached:                                   # the name blob/offsets handed to the sink agree with the list
            assert [b.names_blob[b.name_off[i]:b.name_off[i + 1]].decode() for i in range(len(b.names))] == b.names
    with pytest.raises(ValueError):
        list(iter_packed_blocks_cached(p.iloc[::-1], path, 0.01, 0.99, 64))


def test_buffer_stays_bounded_over_many_blocks(tmp_path):
    """The reader releases consumed text: whatever the file size, it buffers at most one block of lines plus one 8 MB read slab (it used
    to keep the whole inflated file; the reference streams line by line, pyseer/input.py:301-454)."""
    samples = ["sample%04d" % i for i in range(400)]
    rng = np.random.default_rng(3)
    path = str(tmp_path / "big.txt")
    nlines = 6000
    with open(path, "w") as fh:
        for v in range(nlines):
            carriers = np.nonzero(rng.random(400) < 0.5)[0]
            fh.write("K%06d | %s\n" % (v, " ".join("%s:1" % samples[i] for i in carriers)))
    size = os.path.getsize(path)
    assert size > 12 << 20                                       # more than one read slab, so a leak would show
    r = NativeKmerReader(path, samples, 100)
    peak, total = 0, 0
    for bits, counts, blob, off in r.raw_blocks():
        total += counts.shape[0]
        peak = max(peak, int(r._lib.sh_reader_buffered(r._h)))
    assert total == nlines
    assert peak <= (8 << 20) + 2 * (100 * (size // nlines + 64)) + (1 << 16), (peak, size)


def test_long_variant_names_grow_the_names_buffer(tmp_path):
    """Unitig input: names of tens of kilobases.  A block whose names exceed the buffer is refused without being consumed (-2) and
    retried with a larger buffer; nothing is lost or reordered."""
    samples = ["s%d" % i for i in range(20)]
    rng = np.random.default_rng(4)
    names = ["".join(rng.choice(list("ACGT"), int(n))) for n in (5, 40000, 7, 90000, 31, 12, 65000)]
    path = str(tmp_path / "unitigs.gz")
    with gzip.open(path, "wt") as fh:
        for i, nm in enumerate(names):
            fh.write("%s | s%d:1 s%d:1\n" % (nm, i, i + 1))
    r = NativeKmerReader(path, samples, 3, max_name=16)          # 48 bytes of names buffer to start with
    got = []
    for nm, bits, counts in r:
        got += nm
        assert (counts == 2).all()
    assert got == names
