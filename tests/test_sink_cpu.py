"""Native result sink (csrc/writer.cpp, sh_format_rows) against the per-row formatter (pyseer_amd/utils.format_output, itself
checked against the reference's logs by the CLI goldens)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def _cols(rng, n, k):
    cols = [10.0 ** rng.uniform(-320, 300, n) * rng.choice([-1, 1], n) for _ in range(k)]
    for c in cols:
        c[rng.random(n) < 0.1] = np.nan
    cols[2][5] = np.inf; cols[1][7] = 0.0; cols[3][8] = 9.995e-5; cols[3][9] = 1.005; cols[3][10] = -0.0
    cols[0][11] = 5e-324; cols[0][12] = 1.7976931348623157e308; cols[4][13] = 0.5; cols[4][14] = 9.995; cols[4][15] = 99949.99999
    return cols


def test_lmm_rows_match_format_output():
    from pyseer_amd.sink import RowFormatter, names_blob
    from pyseer_amd.utils import format_output
    from pyseer_amd.classes import LMM, notes_from_flags
    rng = np.random.default_rng(0)
    n = 12000                                                 # > 4096: exercises the multi-threaded concatenation
    names = ["K%d_%s" % (i, "ACGT" * (i % 7)) for i in range(n)]
    cols = _cols(rng, n, 6)
    flags = rng.integers(0, 512, n).astype(np.uint32) | (rng.integers(0, 4, n).astype(np.uint32) << 16)
    lin = rng.integers(-1, 3, n).astype(np.int32)
    labels = ["a", "bb", "c"]
    blob, off = names_blob(names)
    sel = rng.permutation(n)[:9000]
    txt = RowFormatter(labels).format(blob, off, sel, cols, flags, lineage=lin).decode().split("\n")
    assert txt[-1] == "" and len(txt) == len(sel) + 1
    for r, v in enumerate(sel):
        x = LMM(names[v], None, cols[0][v], cols[1][v], cols[2][v], cols[3][v], cols[4][v], cols[5][v],
                None if lin[v] < 0 else int(lin[v]), [], [], notes_from_flags(flags[v]), False, False)
        assert format_output(x, labels, 'lmm', False) == txt[r]


def test_seer_rows_with_and_without_betas():
    from pyseer_amd.sink import RowFormatter, names_blob
    from pyseer_amd.utils import format_output
    from pyseer_amd.classes import Seer, notes_from_flags
    rng = np.random.default_rng(1)
    n, q = 500, 4
    names = ["v%d" % i for i in range(n)]
    cols = _cols(rng, n, 6)
    betas = rng.normal(size=(n, q)); betas[3, 1] = np.nan
    valid = (rng.random(n) < 0.7).astype(np.uint8)
    flags = rng.integers(0, 512, n).astype(np.uint32)
    blob, off = names_blob(names)
    sel = np.arange(n)
    txt = RowFormatter().format(blob, off, sel, cols, flags, betas, valid).decode().split("\n")
    for v in range(n):
        x = Seer(names[v], None, cols[0][v], cols[1][v], cols[2][v], cols[3][v], cols[4][v], cols[5][v],
                 betas[v] if valid[v] else np.array([]), None, [], [], notes_from_flags(flags[v]), False, False)
        assert format_output(x, None, 'seer', False) == txt[v]
    assert RowFormatter().format(blob, off, np.zeros(0, dtype=np.int64), cols, flags) == b''


def test_small_buffer_is_regrown():
    from pyseer_amd.sink import RowFormatter, names_blob
    import ctypes as C
    names = ["x" * 300 for _ in range(50)]
    blob, off = names_blob(names)
    f = RowFormatter(); f._buf = C.create_string_buffer(16)
    out = f.format(blob, off, np.arange(50), [np.zeros(50)], np.zeros(50, dtype=np.uint32))
    assert out.count(b"\n") == 50 and out.startswith(b"x" * 300 + b"\t0.00E+00\t\n")


def test_native_number_format_equals_printf_on_a_sweep():
    """The sink formats numbers without printf since round 3 (csrc/writer.cpp put_num: scale to [100, 1000) in 80-bit arithmetic, round,
    leave every near-tie to snprintf).  '%.2E' % x is the reference's own formatting (pyseer/utils.py:60-75): random bit patterns, p-value
    like magnitudes down to 1e-300, and the exact x.xx5 ties of short decimals must all agree."""
    from pyseer_amd.sink import RowFormatter, names_blob
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 2 ** 63, 120000, dtype=np.int64).view(np.float64) * rng.choice([-1.0, 1.0], 120000)
    raw = raw[np.isfinite(raw)]
    mags = 10.0 ** (-300 * rng.random(60000)) * rng.choice([-1.0, 1.0], 60000)
    ties = np.array([(m + 0.5) * 10.0 ** (e - 2) for m in range(100, 1000, 7) for e in range(-12, 13)])
    ties = np.concatenate([ties, np.nextafter(ties, 0), np.nextafter(ties, np.inf)])
    special = np.array([0.0, -0.0, 1.125, 1.135, 9.995, 999.5, 99.95, 0.125, 5e-324, 1.7976931348623157e308, 2.2250738585072014e-308, 1e22, 1e23, 1e-23])
    x = np.concatenate([raw, mags, ties, special])
    n = x.shape[0]
    blob, off = names_blob(["v"] * n)
    txt = RowFormatter().format(blob, off, np.arange(n), [x], np.zeros(n, dtype=np.uint32)).decode().split("\n")
    got = [t.split("\t")[1] for t in txt[:n]]
    want = ["%.2E" % v for v in x]
    bad = [(v, g, w) for v, g, w in zip(x, got, want) if g != w]
    assert not bad, bad[:5]


def test_a_smaller_openmp_team_than_asked_for_loses_nothing():
    """the formatter cuts the rows into as many parts as it ASKS threads for; a runtime that grants fewer (OMP_THREAD_LIMIT, nested
    regions) must still fill every part, and a part must never keep the text of an earlier call.  Two calls of different length under
    OMP_THREAD_LIMIT=2 in a child process against the same calls here."""
    import subprocess
    code = r'''
import sys, numpy as np, hashlib
sys.path.insert(0, %r)
from pyseer_amd.sink import RowFormatter, names_blob
rng = np.random.default_rng(5)
f = RowFormatter()
for n in (30000, 9000):
    names = ["k%%d" %% i for i in range(n)]
    cols = [rng.normal(size=n) for _ in range(6)]
    flags = rng.integers(0, 512, n).astype(np.uint32)
    blob, off = names_blob(names)
    t = f.format(blob, off, np.arange(n), cols, flags)
    print(len(t), hashlib.md5(t).hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for lim in (None, "2", "1"):
        env = dict(os.environ)
        env.pop("OMP_THREAD_LIMIT", None)
        if lim:
            env["OMP_THREAD_LIMIT"] = lim
        r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] and len(outs[0].splitlines()) == 2


def test_two_sinks_format_concurrently():
    """Round 4 held a process-wide mutex for the whole of sh_format_rows (one shared array of text buffers), so the sinks of a --gpus job
    formatted one after the other.  Now every calling thread owns its buffers: two threads are inside the formatter at the same time
    (sh_format_concurrency_max) and each gets exactly the text it gets alone."""
    import threading
    from pyseer_amd import _abi
    from pyseer_amd.sink import RowFormatter, names_blob
    lib = _abi.load()
    rng = np.random.default_rng(3)
    n = 300000
    names = ["K%d" % i for i in range(n)]
    blob, off = names_blob(names)
    jobs = []
    for s in range(2):
        cols = [rng.normal(size=n) * 10.0 ** rng.integers(-30, 30, n) for _ in range(6)]
        flags = rng.integers(0, 512, n).astype(np.uint32)
        jobs.append((cols, flags, np.arange(n, dtype=np.int64)))
    alone = [RowFormatter().format(blob, off, sel, cols, flags) for cols, flags, sel in jobs]
    lib.sh_format_concurrency_max(1)
    out = [None, None]
    go = threading.Barrier(2)

    def work(i):
        f = RowFormatter()
        cols, flags, sel = jobs[i]
        go.wait()
        for _ in range(6):
            out[i] = f.format(blob, off, sel, cols, flags)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert out[0] == alone[0] and out[1] == alone[1]
    assert lib.sh_format_concurrency_max(0) == 2
    assert lib.sh_host_pool_workers() <= max(0, lib.sh_host_cpus() - 1)      # one persistent pool, never more workers than the budget


def test_format_records_equals_format_rows():
    """sh_format_records (the job stream's sink: compacted records, names and af by variant index) prints what sh_format_rows prints for the
    same rows with the af column made by the caller."""
    import ctypes as C
    from pyseer_amd import _abi
    from pyseer_amd.sink import RowFormatter, names_blob
    lib = _abi.load()
    rng = np.random.default_rng(4)
    nv, n_samples, q = 20000, 777, 3
    names = ["V%d%s" % (i, "T" * (i % 5)) for i in range(nv)]
    blob, off = names_blob(names)
    counts = rng.integers(0, n_samples + 1, nv).astype(np.int32)
    idx = np.sort(rng.choice(nv, 9000, replace=False)).astype(np.int32)
    nsel = idx.shape[0]; cap = nsel + 37
    cols = rng.normal(size=(5 + q, cap)) * 10.0 ** rng.integers(-200, 200, (5 + q, cap)); cols[1, ::7] = np.nan; cols[2, ::7] = np.nan
    flags = rng.integers(0, 512, nsel).astype(np.uint32)
    valid = (np.isfinite(cols[2, :nsel]) | np.isfinite(cols[1, :nsel])).astype(np.uint8)
    cp = (_abi.c_dp * 5)(*[cols[a].ctypes.data_as(_abi.c_dp) for a in range(5)])
    text = C.c_void_p()
    n = lib.sh_format_records(blob, off.ctypes.data_as(C.POINTER(C.c_int64)), counts.ctypes.data_as(C.POINTER(C.c_int32)), n_samples,
                              idx.ctypes.data_as(C.POINTER(C.c_int32)), nsel, cp, 5, cols[5].ctypes.data_as(_abi.c_dp), cap, q,
                              valid.ctypes.data_as(_abi.c_u8p), None, None, 0, flags.ctypes.data_as(_abi.c_u32p), C.byref(text))
    got = C.string_at(text.value, n)
    # ... and with the lineage column (round 6: sh_job_set_lineage): labels by index, NA for -1
    labels = ["MDS3", "cluster_B", "x"]
    lin = rng.integers(-1, 3, nsel).astype(np.int32)
    lab = (C.c_char_p * 3)(*[x.encode() for x in labels])
    n2 = lib.sh_format_records(blob, off.ctypes.data_as(C.POINTER(C.c_int64)), counts.ctypes.data_as(C.POINTER(C.c_int32)), n_samples,
                               idx.ctypes.data_as(C.POINTER(C.c_int32)), nsel, cp, 5, cols[5].ctypes.data_as(_abi.c_dp), cap, q,
                               valid.ctypes.data_as(_abi.c_u8p), lin.ctypes.data, C.cast(lab, C.c_void_p), 3, flags.ctypes.data_as(_abi.c_u32p), C.byref(text))
    got_lin = C.string_at(text.value, n2)
    # the same through sh_format_rows: values scattered back to variant positions
    full = [np.full(nv, np.nan) for _ in range(6)]
    full[0] = counts.astype(np.float64) / n_samples
    for a in range(5):
        full[a + 1][idx] = cols[a, :nsel]
    fb = np.full((nv, q), np.nan); fb[idx] = cols[5:, :nsel].T
    fv = np.zeros(nv, np.uint8); fv[idx] = valid
    ff = np.zeros(nv, np.uint32); ff[idx] = flags
    want = RowFormatter().format(blob, off, idx.astype(np.int64), full, ff, fb, fv)
    assert got == want
    fl = np.full(nv, -1, np.int32); fl[idx] = lin
    assert got_lin == RowFormatter(labels).format(blob, off, idx.astype(np.int64), full, ff, fb, fv, fl)
    assert got_lin != got and got_lin.count(b"\tNA\t") >= int((lin < 0).sum())


def test_the_gather_buffer_can_be_handed_over_and_replaced():
    """csrc/writer.cpp format_gather_detach / format_gather_attach: the block loop's writer thread (csrc/job_run.inc OutWriter) takes the formatter's
    gathered text as the buffer itself and gives a finished buffer back.  Detaching anything but the calling thread's gather buffer is refused;
    after a detach the next call gathers into a fresh buffer, after an attach into the one given -- the same text every time."""
    import ctypes as C
    from pyseer_amd import _abi
    from pyseer_amd.sink import names_blob
    lib = _abi.load()
    lib.format_gather_detach.restype = C.c_void_p; lib.format_gather_detach.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.format_gather_attach.restype = None; lib.format_gather_attach.argtypes = [C.c_void_p, C.c_size_t]
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(9)
    nv, n_samples = 30000, 500
    blob, off = names_blob(["K%d" % i for i in range(nv)])
    counts = rng.integers(0, n_samples + 1, nv).astype(np.int32)
    idx = np.arange(nv, dtype=np.int32)
    cols = rng.normal(size=(5, nv))
    flags = np.zeros(nv, np.uint32)
    cp = (_abi.c_dp * 5)(*[cols[a].ctypes.data_as(_abi.c_dp) for a in range(5)])

    def fmt():
        text = C.c_void_p()
        n = lib.sh_format_records(blob, off.ctypes.data_as(C.POINTER(C.c_int64)), counts.ctypes.data_as(C.POINTER(C.c_int32)), n_samples,
                                  idx.ctypes.data_as(C.POINTER(C.c_int32)), nv, cp, 5, None, 0, 0, None, None, None, 0,
                                  flags.ctypes.data_as(_abi.c_u32p), C.byref(text))
        return text.value, n
    if lib.sh_host_cpus() < 2:
        import pytest
        pytest.skip("one CPU: the text is the one part itself, nothing is gathered")
    t0, n0 = fmt()
    want = C.string_at(t0, n0)
    assert n0 > (1 << 20)
    cap = C.c_size_t(0)
    assert lib.format_gather_detach(t0 + 1, C.byref(cap)) is None            # not the gather buffer
    p = lib.format_gather_detach(t0, C.byref(cap))
    assert p == t0 and cap.value >= n0
    assert lib.format_gather_detach(t0, C.byref(cap)) is None                # already gone
    t1, n1 = fmt()                                                          # a fresh buffer; the detached one is still ours
    assert C.string_at(t1, n1) == want and C.string_at(p, n0) == want
    if lib.sh_host_cpus() > 1:
        assert t1 != p
    lib.format_gather_attach(p, cap)                                        # (frees the fresh buffer, gathers into the old one from now on)
    t2, n2 = fmt()
    assert t2 == p and C.string_at(t2, n2) == want
