"""Native k-mer reader (pyseer_amd/csrc/reader.cpp) against the Python restatement of pyseer/input.py:read_variant, on the
reference's own kmers.gz and on synthetic gz / plain files with the format's corner cases."""
import gzip
import os

import numpy as np
import pytest
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
        # the native path keeps arrays and hands every parsed row over in place (the engine applies the AF window itself); the Python
        # restatement compacts the kept rows as the reference's iter_variants drops the filtered ones
        assert a.names == b.names and list(a.status) == list(b.status)
        kept = np.asarray(a.status) == 0
        assert np.array_equal(np.asarray(a.bits)[np.asarray(a.row_of)[kept]], np.asarray(b.bits)[np.asarray(b.row_of, dtype=np.int64)[kept]])
        assert np.allclose(a.afs, b.afs)
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


def test_tokens_of_every_length_against_the_python_parser(tmp_path):
    """The parser's 16-byte steps (token ends, 16-byte name keys; longer names through the string compare): sample names of 1 to 40
    characters, names that agree in their first 16 bytes, a name that is a prefix of another, tokens with and without `:value`, with
    several colons, unknown names of every length, tabs / CR / runs of blanks, the last token right at the end of a file without a
    newline -- the presence vectors of read_variant (pyseer/input.py:377-388, restated in pyseer_amd/input.py)."""
    rng = np.random.default_rng(11)
    samples = []
    for n in range(1, 41):
        samples.append("".join(rng.choice(list("abcdefgh0123_."), n)))
    samples += ["sample_0123456789", "sample_0123456789x", "sample_0123456789xy", "sample_012345678", "sample_01234567", "p", "pp", "ppp"]
    samples = list(dict.fromkeys(samples))
    unknown = ["".join(rng.choice(list("QRSTUV"), n)) for n in (1, 7, 8, 9, 15, 16, 17, 31, 32, 33)] + [samples[20] + "Z", samples[30][:-1]]
    seps = [" ", "  ", "\t", " \t ", "   "]
    lines = []
    for v in range(300):
        toks = []
        for name in rng.permutation(samples + unknown)[:rng.integers(0, len(samples) + len(unknown) + 1)]:
            form = rng.integers(0, 4)
            toks.append(name if form == 0 else name + ":1" if form == 1 else name + ":17:3" if form == 2 else name + ":")
        body = "".join(rng.choice(seps) + t for t in toks)
        lines.append("K%04d |%s%s" % (v, body, rng.choice(["", " ", "\r", " \r"])))
    p = pd.Series(np.arange(len(samples), dtype=float), index=samples)
    for name, opener in (("t.txt", open), ("t.gz", gzip.open)):
        path = str(tmp_path / name)
        with opener(path, "wt", newline="") as fh:
            fh.write("\n".join(lines))                           # no trailing newline
        got = [(n, b.copy(), c.copy()) for n, b, c in NativeKmerReader(path, samples, 64)]
        names = sum((g[0] for g in got), [])
        bits = np.concatenate([g[1] for g in got]); counts = np.concatenate([g[2] for g in got])
        assert names == ["K%04d" % v for v in range(300)]
        fh = open(path, newline="") if name.endswith(".txt") else gzip.open(path, "r")
        for v in range(300):
            eof, k, var_name, ks, nks, af, missing = read_variant(fh, p, "kmers", False, None, name.endswith(".txt"), set(samples), [])
            dense = np.unpackbits(bits[v], bitorder="little")[:len(samples)]
            assert var_name == names[v] and np.array_equal(dense, k), (v, lines[v])
            assert counts[v] == int(np.sum(k))


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


def test_buffer_stays_bounded_over_many_blocks(tmp_path, monkeypatch):
    """The reader holds a fixed number of recycled slab buffers whatever the file size (round 1 kept the whole inflated file; the
    reference streams line by line, pyseer/input.py:301-454): ~50 MB of text through 1 MB slabs never allocates more than 16 of them."""
    samples = ["sample%04d" % i for i in range(400)]
    rng = np.random.default_rng(3)
    nlines = 24000
    lines = []
    for v in range(nlines):
        carriers = np.nonzero(rng.random(400) < 0.5)[0]
        lines.append("K%06d | %s\n" % (v, " ".join("%s:1" % samples[i] for i in carriers)))
    text = "".join(lines).encode()
    assert len(text) > 40 << 20
    monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=%d,reader_pad=%d" % (1 << 20, 1 << 18))
    for name, blob in (("big.gz", gzip.compress(text, 1)), ("big.txt", text)):
        path = str(tmp_path / name)
        open(path, "wb").write(blob)
        r = NativeKmerReader(path, samples, 100)
        peak, total = 0, 0
        for bits, counts, blob2, off in r.raw_blocks():
            total += counts.shape[0]
            peak = max(peak, int(r._lib.sh_reader_buffered(r._h)))
        assert total == nlines
        assert peak <= 16 * ((1 << 20) + (1 << 18)), (name, peak)         # look-ahead of 10 slabs + the block in hand + the one being filled


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


def _bgzf(data, member=40000):
    """BGZF as bgzip writes it: independent gzip members of <= 64 KB, each with the BC extra subfield (its compressed size - 1)."""
    import struct
    import zlib
    out = b""
    for i in range(0, max(len(data), 1), member):
        chunk = data[i:i + member]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(comp) + 8
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk) & 0xFFFFFFFF)
    return out + b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00"


def _kmer_text(samples, nlines, seed):
    rng = np.random.default_rng(seed)
    rows = []
    for v in range(nlines):
        carriers = np.nonzero(rng.random(len(samples)) < rng.uniform(0.05, 0.95))[0]
        rows.append("K%05d | %s" % (v, " ".join("%s:%d" % (samples[i], 1 + i % 3) for i in carriers)))
    return ("\n".join(rows) + "\n").encode()


def test_every_container_gives_the_same_blocks(tmp_path, monkeypatch):
    """Plain text, single-member gzip (levels 1/6/9, stored blocks), concatenated members, BGZF, and zlib's own gzread as the yardstick:
    the in-tree inflate (csrc/inflate_fast.h) on one thread (SEERHIP_ROUTE reader=serial) and on several (csrc/inflate_par.h; by default only for
    files of two chunks or more, so the chunk is made tiny here) and the member-parallel BGZF path hand the parser the same bytes."""
    import zlib
    samples = ["iso%03d" % i for i in range(150)]
    text = _kmer_text(samples, 1200, 5)
    files = {"plain.txt": text, "l1.gz": gzip.compress(text, 1), "l6.gz": gzip.compress(text, 6), "l9.gz": gzip.compress(text, 9),
             "stored.gz": gzip.compress(text, 0),
             "multi.gz": gzip.compress(text[:70001], 6) + gzip.compress(b"", 6) + gzip.compress(text[70001:], 9),
             "bgzf.gz": _bgzf(text)}
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    files["fixed.gz"] = co.compress(text) + co.flush()
    monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=50000")             # many slabs, member and block boundaries inside them
    want = None
    for name, blob in files.items():
        path = str(tmp_path / name)
        open(path, "wb").write(blob)
        for sel in ("", "zlib", "serial", "par4096", "par20000"):
            if sel:
                if name == "plain.txt":
                    continue
                if sel.startswith("par"):                                   # one member on several threads (csrc/inflate_par.h), tiny chunks
                    monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=50000,reader_chunk=" + sel[3:])
                else:
                    monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=50000,reader=" + sel)
            got = [(n, b.copy(), c.copy()) for n, b, c in NativeKmerReader(path, samples, 97)]
            monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=50000")
            names = sum((g[0] for g in got), [])
            bits = np.concatenate([g[1] for g in got]); counts = np.concatenate([g[2] for g in got])
            if want is None:
                want = (names, bits, counts)
                assert len(names) == 1200
            assert names == want[0] and np.array_equal(bits, want[1]) and np.array_equal(counts, want[2]), (name, sel)


def test_one_gzip_member_on_several_threads(tmp_path, monkeypatch):
    """csrc/inflate_par.h (block heads searched for, chunks decoded to symbols with window markers, the chain accepted in order): the rows
    of the one-thread decoder, for compression levels 1/6/9, several members, a member with stored and fixed-Huffman blocks between dynamic
    ones, chunk sizes below and above a block, 3 and 8 reader threads; the several-thread path did run (sh_reader_par_chunks); a flipped
    bit anywhere is an error, never a silently different block.  Reference: one gzip.open stream, pyseer/input.py:271-276."""
    import zlib, pytest
    samples = ["iso%03d" % i for i in range(150)]
    text = _kmer_text(samples, 7000, 6)
    cut = [len(text) // 5, 2 * len(text) // 5, 3 * len(text) // 5]
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    mixed = co.compress(text[:cut[0]]) + co.flush(zlib.Z_FULL_FLUSH)
    co2 = zlib.compressobj(0, zlib.DEFLATED, -15)                      # the same member continued by hand: stored, fixed, dynamic again
    files = {"l1.gz": gzip.compress(text, 1), "l6.gz": gzip.compress(text, 6), "l9.gz": gzip.compress(text, 9),
             "multi.gz": gzip.compress(text[:cut[1]], 6) + gzip.compress(b"", 6) + gzip.compress(text[cut[1]:], 9)}
    def raw(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, last=False):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        out = c.compress(data)
        return out + (c.flush() if last else c.flush(zlib.Z_FULL_FLUSH))
    body = (raw(text[:cut[0]], 6) + raw(text[cut[0]:cut[0] + 70000], 0) + raw(text[cut[0] + 70000:cut[1]], 6, zlib.Z_FIXED)
            + raw(text[cut[1]:], 6, last=True))
    files["mixed.gz"] = (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + body
                         + (zlib.crc32(text) & 0xFFFFFFFF).to_bytes(4, "little") + (len(text) & 0xFFFFFFFF).to_bytes(4, "little"))
    assert gzip.decompress(files["mixed.gz"]) == text
    del co, co2, mixed
    def read(path, route, sel=None, threads=None):
        monkeypatch.setenv("SEERHIP_ROUTE", ",".join([route] + ["%s=%s" % (k, v) for k, v in (("reader", sel), ("reader_threads", threads)) if v is not None]))
        rd = NativeKmerReader(path, samples, 1000)
        names, bits, counts, pc = [], [], [], 0
        for n, b, c in rd:
            names += n; bits.append(b.copy()); counts.append(c.copy())
            if rd._h: pc = int(rd._lib.sh_reader_par_chunks(rd._h))
        return names, np.concatenate(bits), np.concatenate(counts), pc
    seen_small = False
    for name, blob in files.items():
        path = str(tmp_path / name)
        open(path, "wb").write(blob)
        want = read(path, "reader_slab=300000", "serial")
        assert len(want[0]) == 7000 and want[3] == 0
        for chunk in (8192, 40000):
            for threads in ("3", "8"):
                got = read(path, "reader_slab=300000,reader_chunk=%d" % chunk, None, threads)
                assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (name, chunk, threads)
                assert got[3] > 0 or chunk == 8192, (name, chunk, threads)      # (a chunk smaller than a block may hold no block head)
                seen_small = seen_small or (chunk == 8192 and got[3] > 0)
    assert seen_small
    # the decoder's sizing keys (csrc/route.h): few decoding threads, one translating helper, a look-ahead of two slabs, regions sized for
    # 200 KB of text -- other schedules of the same work, the same rows
    path = str(tmp_path / "l6.gz")
    want = read(path, "reader_slab=300000", "serial")
    got = read(path, "reader_slab=300000,reader_chunk=40000,reader_workers=2,reader_helpers=1,reader_depth=2,reader_target=200000", None, "6")
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and got[3] > 0
    # corruption: a flipped bit early, in the middle, near the end, in the CRC; truncation
    good = files["l6.gz"]
    bad = [good[:k] + bytes([good[k] ^ 0x10]) + good[k + 1:] for k in (len(good) // 7, len(good) // 2, len(good) - 3000, len(good) - 6)]
    bad.append(good[:len(good) - 40])
    for k, blob in enumerate(bad):
        path = str(tmp_path / ("bad%d.gz" % k))
        open(path, "wb").write(blob)
        with pytest.raises(IOError):
            read(path, "reader_slab=300000,reader_chunk=8192", None, "8")


def test_clmul_crc32_equals_zlib():
    """csrc/crc32_clmul.h (carry-less-multiplication folding of the gzip CRC-32; zlib's routine cost as much host time as inflating):
    the value of zlib.crc32 for every length around the 16- and 64-byte block sizes, random starts and initial values."""
    import ctypes as C
    import zlib
    from pyseer_amd import _abi
    lib = C.CDLL(_abi.LIB_PATH)
    lib.shk_crc32.restype = C.c_uint32
    lib.shk_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_int64]
    rng = np.random.default_rng(8)
    data = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    for n in list(range(0, 300)) + [4095, 4096, 4097, 65535, 65536, 65537, 1 << 20] + [int(x) for x in rng.integers(300, 1 << 20, 200)]:
        off = int(rng.integers(0, len(data) - n + 1))
        init = int(rng.integers(0, 1 << 32)) if n % 3 else 0
        assert lib.shk_crc32(init, data[off:off + n], n) == zlib.crc32(data[off:off + n], init), n


def test_corrupt_gzip_is_reported(tmp_path):
    """A flipped bit inside the deflate data: the stream either stops decoding or fails its CRC-32 -- never a silent wrong block."""
    import pytest
    samples = ["iso%03d" % i for i in range(60)]
    text = _kmer_text(samples, 400, 9)
    good = gzip.compress(text, 6)
    for k, blob in enumerate((good[:len(good) // 2] + bytes([good[len(good) // 2] ^ 0x10]) + good[len(good) // 2 + 1:],
                              good[:-6] + bytes([good[-6] ^ 1]) + good[-5:],          # CRC field itself
                              good[:len(good) - 40],                                  # truncated
                              _bgzf(text)[:300] + b"\x00" + _bgzf(text)[301:])):
        path = str(tmp_path / ("bad%d.gz" % k))
        open(path, "wb").write(blob)
        with pytest.raises(IOError):
            for _ in NativeKmerReader(path, samples, 50):
                pass


def test_lines_longer_than_the_pad_and_blocks_spanning_many_slabs(tmp_path, monkeypatch):
    """Slab geometry must not show: lines longer than the pad (copied across the boundary), blocks that span many slabs, a file without
    a trailing newline -- same rows as with one big slab."""
    samples = ["iso%03d" % i for i in range(300)]
    text = _kmer_text(samples, 700, 21)[:-1]                       # no trailing newline
    want = None
    for slab, pad, bs in ((1 << 24, 1 << 20, 97), (70000, 32768, 97), (70000, 32768, 5000), (200000, 32768, 1)):
        monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=%d,reader_pad=%d" % (slab, pad))
        for name, blob in (("t.txt", text), ("t.gz", gzip.compress(text, 6)), ("t.bgzf.gz", _bgzf(text))):
            path = str(tmp_path / name)
            open(path, "wb").write(blob)
            got = [(n, b.copy(), c.copy()) for n, b, c in NativeKmerReader(path, samples, bs)]
            names = sum((g[0] for g in got), [])
            bits = np.concatenate([g[1] for g in got]); counts = np.concatenate([g[2] for g in got])
            if want is None:
                want = (names, bits, counts); assert len(names) == 700
            assert names == want[0] and np.array_equal(bits, want[1]) and np.array_equal(counts, want[2]), (slab, pad, bs, name)
    # lines of ~45 KB against a 32 KB pad: the bridge
    samples2 = ["s%05d" % i for i in range(6000)]
    text2 = _kmer_text(samples2, 60, 22)
    monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=100000,reader_pad=32768")
    ref = None
    for name, blob in (("u.txt", text2), ("u.gz", gzip.compress(text2, 6))):
        path = str(tmp_path / name); open(path, "wb").write(blob)
        got = [(n, b.copy(), c.copy()) for n, b, c in NativeKmerReader(path, samples2, 7)]
        cur = (sum((g[0] for g in got), []), np.concatenate([g[1] for g in got]))
        if ref is None:
            ref = cur; assert len(cur[0]) == 60
        assert cur[0] == ref[0] and np.array_equal(cur[1], ref[1])
    monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=%d,reader_pad=%d" % (1 << 24, 1 << 20))
    big = [(n, b.copy()) for n, b, c in NativeKmerReader(str(tmp_path / "u.gz"), samples2, 7)]
    assert sum((g[0] for g in big), []) == ref[0] and np.array_equal(np.concatenate([g[1] for g in big]), ref[1])


def test_several_files_are_one_stream_in_the_order_given(tmp_path):
    """`--kmers a.gz b.txt.gz c.gz`: every file has its own reader thread (one gzip stream inflates serially), the consumer drains them
    in the order given.  Same variants, same order, same rows as the concatenated file; a small read-ahead budget makes the readers of
    the later files wait without changing anything; an unreadable file is reported when its turn comes."""
    import pytest
    from pyseer_amd.input import iter_packed_blocks_native_multi
    samples = ["s%03d" % i for i in range(150)]
    p = pd.Series(np.arange(len(samples), dtype=float), index=samples)
    texts = [_kmer_text(samples, n, seed) if n else b"" for n, seed in ((700, 1), (3, 2), (1500, 3), (0, 4), (260, 5))]
    paths = []
    for i, t in enumerate(texts):
        path = str(tmp_path / ("part%d.gz" % i))
        with open(path, "wb") as fh:
            fh.write(_bgzf(t) if i == 2 else gzip.compress(t))
        paths.append(path)
    whole = str(tmp_path / "whole.gz")
    with open(whole, "wb") as fh:
        fh.write(gzip.compress(b"".join(texts)))

    def flat(blocks):
        names, rows, st = [], [], []
        for b in blocks:
            names += list(b.names); st += list(b.status)
            full = np.zeros((len(b.names), b.bits.shape[1]), dtype=np.uint8)
            on = [i for i, s in enumerate(b.status) if s == 0]
            full[on] = b.bits[[b.row_of[i] for i in on]]
            rows.append(full)
        return names, st, (np.concatenate(rows) if rows else np.zeros((0, 0), dtype=np.uint8))
    want = flat(iter_packed_blocks_native(p, whole, 0.01, 0.99, 64))
    assert len(want[0]) == 700 + 3 + 1500 + 260
    for budget in (2 << 30, 1):                              # 1 byte: at most one block of a later file is parsed ahead
        got = flat(iter_packed_blocks_native_multi(p, paths, 0.01, 0.99, 64, ahead_bytes=budget))
        assert got[0] == want[0] and got[1] == want[1] and np.array_equal(got[2], want[2])
    with pytest.raises(IOError):
        list(iter_packed_blocks_native_multi(p, [paths[0], str(tmp_path / "missing.gz"), paths[4]], 0.01, 0.99, 64))


def test_block_larger_than_any_fixed_number_of_slabs_does_not_hang(tmp_path, monkeypatch):
    """sh_reader_next holds every slab of the block it is cutting, and the producer's buffer pool used to be capped at
    64: a block spanning more slabs than that left producer and consumer waiting for each other.  Smallest slabs, one block for the whole
    file (gzip and BGZF-free plain gzip mode both go through get_buf); run in a child process so that a hang fails the test instead of the
    suite."""
    import subprocess
    import sys
    samples = ["s%d" % i for i in range(300)]
    rng = np.random.default_rng(5)
    lines = []
    for v in range(9000):
        car = np.flatnonzero(rng.random(300) < 0.5)
        lines.append("K%06d | %s" % (v, " ".join("s%d:1" % i for i in car)))
    path = str(tmp_path / "big.gz")
    with gzip.open(path, "wt", compresslevel=1) as fh:
        fh.write("\n".join(lines) + "\n")
    text_bytes = sum(len(x) + 1 for x in lines)
    assert text_bytes > 80 * 70000                                 # more than 64 of the smallest slabs in ONE block
    code = ("import sys, numpy as np\n"
            "from pyseer_amd.input import NativeKmerReader\n"
            "s = ['s%d' % i for i in range(300)]\n"
            "n = 0; c = 0\n"
            "for names, bits, counts in NativeKmerReader(sys.argv[1], s, 10000):\n"
            "    n += len(names); c += int(counts.sum())\n"
            "print(n, c)\n")
    env = dict(os.environ, SEERHIP_ROUTE="reader_slab=70000,reader_pad=32768",
               PYTHONPATH=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    out = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    n, c = (int(x) for x in out.stdout.split())
    assert n == 9000 and c == sum(x.count(":1") for x in lines)


def test_trailing_garbage_after_the_last_gzip_member_is_an_error_zero_padding_is_not(tmp_path):
    samples = ["s0", "s1", "s2"]
    body = gzip.compress(b"AAAA | s0:1 s2:1\nCCCC | s1:1\n")
    ok = str(tmp_path / "padded.gz"); bad = str(tmp_path / "garbage.gz")
    open(ok, "wb").write(body + b"\0" * 700)
    open(bad, "wb").write(body + b"not gzip at all")
    got = [n for n, b, c in NativeKmerReader(ok, samples, 10)]
    assert sum(got, []) == ["AAAA", "CCCC"]
    try:
        list(NativeKmerReader(bad, samples, 10))
        raise AssertionError("trailing garbage went unnoticed")
    except (IOError, OSError, ValueError, RuntimeError) as ex:
        assert "garbage" in str(ex) or "gzip" in str(ex)


def test_packed_cache_is_written_atomically_and_truncation_is_reported(tmp_path):
    """the cache appears under its name only when close() has written the terminator; an interrupted writer leaves
    nothing; a truncated file (older version, disk full) is reported as such, not as a reshape error."""
    from pyseer_amd.input import PackedCacheWriter, iter_packed_blocks_cached, packed_cache_complete, row_bytes_for
    samples = ["s%d" % i for i in range(20)]
    p = pd.Series(np.arange(20, dtype=float), index=samples)
    rb = row_bytes_for(20)
    path = str(tmp_path / "k.seerpack")
    stamp = (path + ".stamp", "123 456")
    w = PackedCacheWriter(path, samples, stamp=stamp)
    blob = b"AAAACCCC"; off = np.array([0, 4, 8], dtype=np.int64)
    bits = np.zeros((2, rb), dtype=np.uint8); bits[0, 0] = 3; bits[1, 1] = 1
    w.write_block(blob, off, np.array([2, 1], dtype=np.int32), bits)
    assert not os.path.exists(path) and not os.path.exists(stamp[0])       # nothing under the final names yet
    w.abort()
    assert not os.path.exists(path) and not os.path.exists(path + ".part") and not os.path.exists(stamp[0])
    w = PackedCacheWriter(path, samples, stamp=stamp)
    w.write_block(blob, off, np.array([2, 1], dtype=np.int32), bits)
    w.close()
    assert os.path.exists(path) and open(stamp[0]).read().strip() == "123 456" and packed_cache_complete(path)
    blocks = list(iter_packed_blocks_cached(p, path, 0.0, 1.0, 100))
    assert sum(len(b.names) for b in blocks) == 2
    data = open(path, "rb").read()
    cut = str(tmp_path / "cut.seerpack")
    open(cut, "wb").write(data[:-30])
    assert not packed_cache_complete(cut)
    try:
        list(iter_packed_blocks_cached(p, cut, 0.0, 1.0, 100))
        raise AssertionError("truncation went unnoticed")
    except IOError as ex:
        assert "truncated" in str(ex)


@pytest.mark.parametrize("n_parts", [1, 2, 3, 8])
def test_packed_cache_ranges_partition_the_rows(tmp_path, n_parts):
    """part = (i, n) of iter_packed_blocks_cached (the multi-GPU job: one contiguous range of the cache's rows per device): the ranges are
    disjoint, in order, made of whole blocks of the single stream, and together they are that stream -- names, counts and bits."""
    import pandas as pd
    from pyseer_amd.input import PackedCacheWriter, iter_packed_blocks_cached, row_bytes_for
    from pyseer_amd.packing import pack_variants
    n = 37
    samples = ["s%02d" % i for i in range(n)]
    p = pd.Series(np.arange(n, dtype=float), index=samples)
    rb = row_bytes_for(n)
    rng = np.random.default_rng(12)
    path = str(tmp_path / "c.seerpack")
    w = PackedCacheWriter(path, samples)
    names_all, bits_all = [], []
    for nv in (5, 1, 40, 17, 3, 64):                         # stored blocks of uneven length
        dense = (rng.random((nv, n)) < 0.4).astype(np.uint8)
        bits = pack_variants(dense)
        assert bits.shape[1] == rb
        counts = dense.sum(axis=1).astype(np.int32)
        names = [("k%d_%d" % (len(names_all) + j, j) * (1 + j % 3)).encode() for j in range(nv)]
        off = np.concatenate([[0], np.cumsum([len(x) for x in names])]).astype(np.int64)
        w.write_block(b"".join(names), off, counts, bits)
        names_all += names; bits_all.append(bits)
    w.close()
    bits_all = np.concatenate(bits_all, axis=0)
    single = [np.array(b.bits) for b in iter_packed_blocks_cached(p, path, 0.0, 1.0, 16)]
    got_names, got_blocks = [], []
    for i in range(n_parts):
        for blk in iter_packed_blocks_cached(p, path, 0.0, 1.0, 16, part=(i, n_parts)):
            nv = blk.bits.shape[0]
            got_blocks.append(np.array(blk.bits))
            got_names += [bytes(blk.names_blob[int(blk.name_off[j]):int(blk.name_off[j + 1])]) for j in range(nv)]
            assert int(blk.name_off[0]) == 0 and len(blk.afs) == nv
    # the parts hand over exactly the blocks of the single stream, in order
    assert len(got_blocks) == len(single) and all(np.array_equal(a, b) for a, b in zip(got_blocks, single))
    assert got_names == names_all and np.array_equal(np.concatenate(got_blocks, axis=0), bits_all)


def test_parser_worker_count_does_not_change_the_rows(tmp_path, monkeypatch):
    """SEERHIP_ROUTE reader_threads (parser workers of the native reader; default min(48, cores)): one worker and five give the rows of the default."""
    import gzip
    samples = ["s%03d" % i for i in range(61)]
    p = pd.Series(np.arange(61, dtype=float), index=samples)
    rng = np.random.default_rng(8)
    lines = []
    for v in range(700):
        car = [samples[i] for i in np.nonzero(rng.random(61) < 0.3)[0]]
        lines.append("K%d%s | %s\n" % (v, "ACGT"[v % 4] * (v % 9), " ".join("%s:1" % s for s in car)))
    path = str(tmp_path / "k.gz")
    with gzip.open(path, "wt") as f:
        f.writelines(lines)
    got = {}
    for nt in (None, "1", "5"):
        if nt is None:
            monkeypatch.delenv("SEERHIP_ROUTE", raising=False)
        else:
            monkeypatch.setenv("SEERHIP_ROUTE", "reader_threads=" + nt)
        blocks = list(iter_packed_blocks_native(p, path, 0.0, 1.0, 128))
        got[nt] = (np.concatenate([np.array(b.bits) for b in blocks]), b"".join(bytes(b.names_blob) for b in blocks))
    assert np.array_equal(got[None][0], got["1"][0]) and np.array_equal(got[None][0], got["5"][0])
    assert got[None][1] == got["1"][1] == got["5"][1] and got[None][0].shape[0] == 700


def test_an_unknown_route_key_is_refused(tmp_path, monkeypatch):
    """SEERHIP_ROUTE (csrc/route.h) is the one hook the tests force routes with; a key nobody reads is an error, not a silently ignored typo."""
    p = tmp_path / "k.txt"
    p.write_text("AAAC | s0:1 s1:1\n")
    monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=70000,reader_pda=32768")
    with pytest.raises(IOError, match="reader_pda=32768"):
        NativeKmerReader(str(p), ["s0", "s1"])
    monkeypatch.setenv("SEERHIP_ROUTE", "reader_slab=70000,reader_pad=32768")
    r = NativeKmerReader(str(p), ["s0", "s1"]); r.close()


def test_dma_windows_are_page_disjoint_and_share_one_pinning_budget(monkeypatch):
    """pyseer_amd/input.py _DmaWindows: windows of the packed-cache mapping are registered rounded out to pages, so two of them
    must never share one (a block that starts on the previous window's last page is staged instead); and the bytes of finished windows
    that stay pinned are ONE budget for the process: with G streams alive each keeps KEEP / G."""
    from pyseer_amd import input as I
    monkeypatch.setattr(I._DmaWindows, "WINDOW", 6000)
    w = I._DmaWindows.__new__(I._DmaWindows); w._win_of = {}
    ext = [(100, 5000), (5200, 300), (5600, 9000), (14700, 100), (20000, 4000)]          # (offset of the rows, bytes) of five stored blocks
    w.plan(0x10000, ext, 0, 5)
    wins = sorted({id(x): x for x in w._win_of.values()}.values(), key=lambda x: x["lo"])
    assert [(x["lo"] - 0x10000, x["n"], x["blocks"]) for x in wins] == [(100, 5400, 2), (14700, 100, 1), (20000, 4000, 1)]
    assert sorted(w._win_of) == [0, 1, 3, 4]                                             # block 2 starts on window 0's last page: staged
    pages = [(x["lo"] & ~4095, (x["lo"] + x["n"] + 4095) & ~4095) for x in wins]
    assert all(a[1] <= b[0] for a, b in zip(pages, pages[1:]))
    monkeypatch.setattr(I._DmaWindows, "_live", 8)
    assert w._keep() == I._DmaWindows.KEEP // 8
    monkeypatch.setattr(I._DmaWindows, "_live", 0)
    assert w._keep() == I._DmaWindows.KEEP
