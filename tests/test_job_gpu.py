"""The job stream (include/seerhip.h sh_job_*; csrc/job_api.inc, job_kernels.hip): blocks of packed rows in, the text of the printed rows and the
counters out -- against the path it replaces (host result arrays -> mask_like_fit_lmm -> numpy selects -> sh_format_rows), byte for byte, which
the CLI tests (tests/test_cli_gpu.py) in turn hold to the reference's own output (print loops pyseer/__main__.py:571-593, 805-827;
format_output pyseer/utils.py:39-105)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def in_a_process_of_its_own(fn):
    """The tests that register host memory for DMA (hipHostRegister on a buffer / on windows of a mapped cache file) and let it go again run in a
    child interpreter: a session that had done so was twice seen to abort inside the runtime a test or two later ("open at the end
    of round 6") -- in a process of their own they neither leave anything behind in the session nor take it down with them.  The child runs
    the same test by its node id; its failure is this test's failure, with its output."""
    import functools
    import subprocess

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if os.environ.get("SEERHIP_TEST_CHILD") == "1":
            return fn(*args, **kwargs)
        node = os.environ["PYTEST_CURRENT_TEST"].rsplit(" ", 1)[0]
        env = dict(os.environ, SEERHIP_TEST_CHILD="1", PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-m", "pytest", node, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    return wrapper


def _rows(N, V, seed):
    rng = np.random.default_rng(seed)
    af = np.where(rng.random(V) < 0.1, rng.uniform(0.0, 0.02, V), rng.uniform(0.02, 0.98, V))
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    from pyseer_amd.engine import pack_variants
    bits = pack_variants(K)
    counts = K.sum(axis=1).astype(np.int32)
    names = [("K%06d" % v) + "ACGT" * int(rng.integers(0, 6)) for v in range(V)]
    from pyseer_amd.sink import names_blob
    blob, off = names_blob(names)
    return bits, counts, blob, off


def _old_path(e, lmm, bits, counts, blob, off, n, print_filtered, min_af=0.01, max_af=0.99):
    """What pyseer_amd/__main__.py run_stream.sink_block does with a block (the round-4 sink), restated for one block."""
    from pyseer_amd.sink import RowFormatter
    from pyseer_amd.lmm import mask_like_fit_lmm
    r = e.lmm_batch(bits) if lmm else e.glm_batch(bits)
    if lmm:
        r = mask_like_fit_lmm(r)
    afs = counts.astype(np.float64) / n
    on = (afs >= min_af) & (afs <= max_af)
    keys = ("prep", "pvalue", "beta", "bse", "frac_h2") if lmm else ("prep", "pvalue", "kbeta", "bse", "intercept")
    cols = [afs] + [np.where(on, r[k], np.nan) for k in keys]
    flags = np.where(on, r["flags"], np.uint32(1 | (1 << 16))).astype(np.uint32)
    betas = valid = None
    if not lmm and r["betas"].shape[1]:
        betas = r["betas"]; valid = (on & (np.isfinite(r["kbeta"]) | np.isfinite(r["pvalue"]))).astype(np.uint8)
    pf = (flags & (1 << 16)) != 0; ft = (flags & (1 << 17)) != 0
    order = np.arange(len(on))
    if lmm:
        order = np.concatenate([order[pf], order[~pf]])
    show = np.ones(len(on), bool) if print_filtered else (~pf & ~ft)
    sel = order[show[order]]
    text = RowFormatter().format(blob, off, sel, cols, flags, betas, valid) if len(sel) else b""
    return text, (int(pf.sum()), int((~pf).sum()), int(len(sel)))


def _setup(lmm, N, q=3, pret=1.0, lrtt=1.0, seed=5):
    from pyseer_amd.engine import Engine
    rng = np.random.default_rng(seed)
    e = Engine(N); e.set_af_filter(0.01, 0.99)
    if lmm:
        d = np.load(os.path.join(G, "lmm_N300_D3.npz"))
        assert int(d["N"]) == N
        e.lmm_setup(d["U"], d["S"], d["y"], d["covar"], float(d["h2"]), filter_pvalue=pret, lrt_pvalue=lrtt)
    else:
        from pyseer_amd.model import fit_null
        W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
        y = (rng.random(N) < 1 / (1 + np.exp(-(0.2 + W[:, 0])))).astype(float)
        e0 = np.zeros((0, 0))
        e.glm_setup(y, W, False, fit_null(y, W, e0, False).llf, fit_null(y, W, e0, False, firth=True), pret, lrtt)
    return e


@pytest.mark.parametrize("lmm", [True, False])
@pytest.mark.parametrize("pret,lrtt,print_filtered", [(1.0, 1.0, False), (0.5, 0.2, False), (0.5, 0.2, True), (1.0, 1e-3, False)])
def test_job_stream_equals_the_block_sink(lmm, pret, lrtt, print_filtered):
    """Seven blocks of uneven sizes (one row, not a multiple of the selection kernel's 1024-row groups, more blocks than slots) through one
    Job, three in flight: text and counters of every block equal the host sink's, in order."""
    from pyseer_amd.engine import Job
    N = 300
    e = _setup(lmm, N, pret=pret, lrtt=lrtt)
    sizes = [1500, 1, 1024, 3000, 1025, 7, 2047]
    blocks = [_rows(N, v, 100 + i) for i, v in enumerate(sizes)]
    want = [_old_path(e, lmm, *b, N, print_filtered) for b in blocks]
    job = Job(e, lmm, print_filtered)
    got = []
    for i, (bits, counts, blob, off) in enumerate(blocks):
        job.submit(bits, counts, blob, off)
        while job.pending() > 2:
            t, c, _ = job.collect(); got.append((bytes(t), c))
    while job.pending():
        t, c, _ = job.collect(); got.append((bytes(t), c))
    job.close(); e.close()
    assert len(got) == len(want)
    for i, ((t, c), (wt, wc)) in enumerate(zip(got, want)):
        assert c == wc, (i, c, wc)
        assert t == wt, (i, t[:300], wt[:300])
    assert sum(c[2] for _, c in got) > 0
    if print_filtered:
        assert all(c[2] == s for (_, c), s in zip(got, sizes))


_KEPT = []


@in_a_process_of_its_own
def test_job_stream_reads_registered_rows_by_dma():
    """rows_are_dma: the device reads the rows where they lie (sh_host_register on the caller's memory, as the CLI does with the windows of the
    packed-cache mapping); same text as the staged copy."""
    from pyseer_amd.engine import Job
    from pyseer_amd import _abi
    lib = _abi.load()
    N = 300
    e = _setup(True, N)
    bits, counts, blob, off = _rows(N, 5000, 7)
    big = np.zeros((3 << 20,), dtype=np.uint8)                       # a page-aligned window well above the 1 MB the CLI bothers with
    _KEPT.append(big)                                                # (never freed: its address range is not handed out again in this session)
    view = big[:bits.size].reshape(bits.shape); view[:] = bits
    assert lib.sh_host_register(view.ctypes.data, view.nbytes, 0) == 0, lib.sh_last_error()
    job = Job(e, True)
    job.submit(view, counts, blob, off, rows_are_dma=True)
    t1, c1, _ = job.collect(); t1 = bytes(t1)
    job.submit(bits, counts, blob, off, rows_are_dma=False)
    t2, c2, _ = job.collect(); t2 = bytes(t2)
    job.close()
    assert lib.sh_host_unregister(view.ctypes.data) == 0
    e.close()
    assert t1 == t2 and c1 == c2 and c1[2] > 0


def test_job_stream_refuses_a_fourth_block_in_flight_and_bad_shapes():
    from pyseer_amd.engine import Job
    from pyseer_amd import _abi
    N = 300
    e = _setup(True, N)
    job = Job(e, True)
    assert job.depth == 3
    b = _rows(N, 64, 1)
    for _ in range(3):
        job.submit(*b)
    with pytest.raises(_abi.SeerHipError):
        job.submit(*b)
    assert job.pending() == 3
    for _ in range(3):
        job.collect()
    with pytest.raises(_abi.SeerHipError):
        job.collect()
    narrow = np.zeros((4, 8), dtype=np.uint8)                         # 64 bits < 300 samples
    with pytest.raises(_abi.SeerHipError):
        job.submit(narrow, np.zeros(4, np.int32), b"abcd", np.arange(5, dtype=np.int64))
    job.close(); e.close()
    # a fixed-effects job computes its blocks on the context's lanes: its ring is 2 + lanes deep, and what it returns does not depend on them
    texts = {}
    for lanes in (3, 1):
        e = _setup(False, N)
        e.set_lanes(lanes)
        job = Job(e, False)
        assert job.depth == (2 + lanes if lanes > 1 else 3)
        blocks = [_rows(N, 3000, 7 + i) for i in range(7)]
        got = []
        for b in blocks:
            while job.pending() >= job.depth:
                t, c, _ = job.collect(); got.append((bytes(t), c))
            job.submit(*b)
        if lanes > 1:
            assert job.pending() == job.depth
            with pytest.raises(_abi.SeerHipError):
                job.submit(*blocks[0])
        while job.pending():
            t, c, _ = job.collect(); got.append((bytes(t), c))
        texts[lanes] = got
        job.close(); e.close()
    assert texts[3] == texts[1] and sum(c[2] for _, c in texts[3]) > 0


def test_two_contexts_two_jobs_concurrently():
    """Two job streams on two contexts of one device from two host threads (the --gpus job): each returns what it returns alone; their
    formatting overlaps in time (no process-wide lock in csrc/writer.cpp any more)."""
    import threading
    from pyseer_amd.engine import Job
    from pyseer_amd import _abi
    lib = _abi.load()
    N = 300
    blocks = [_rows(N, 20000, 50 + i) for i in range(6)]
    e0 = _setup(True, N)
    want = [_old_path(e0, True, *b, N, True) for b in blocks]
    e0.close()
    lib.sh_format_concurrency_max(1)
    res = {}

    def work(tag):
        e = _setup(True, N)
        job = Job(e, True, True)
        out = []
        for b in blocks:
            job.submit(*b)
            while job.pending() > 2:
                t, c, _ = job.collect(); out.append((bytes(t), c))
        while job.pending():
            t, c, _ = job.collect(); out.append((bytes(t), c))
        job.close(); e.close()
        res[tag] = out
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for tag in (0, 1):
        assert [x[1] for x in res[tag]] == [w[1] for w in want]
        assert [x[0] for x in res[tag]] == [w[0] for w in want]


@pytest.mark.parametrize("lmm,per_variant,nlin", [(False, False, 4), (False, False, 18), (True, False, 4), (True, True, 4), (True, False, 18)])
@pytest.mark.parametrize("print_filtered", [False, True])
def test_job_stream_lineage_column_equals_the_host_path(lmm, per_variant, nlin, print_filtered):
    """sh_job_set_lineage (round 6): fit_lineage_effect (pyseer/model.py:151-199) inside the stream.  Against what pyseer_amd/__main__.py
    run_stream.sink_block does with sh_lineage_batch on the host's side of the block: fixed effects -- every fitted row (model.py:379-382);
    LMM -- the fit of the block's LAST row for every row that passed (pyseer/lmm.py:209-213, the stale `k`), or each passing row's own with
    the per-variant extension.  nlin = 18: 1 + lineages + covariates > 16, the workgroup-per-variant kernel."""
    from pyseer_amd.engine import Job
    from pyseer_amd.sink import RowFormatter
    from pyseer_amd.lmm import mask_like_fit_lmm
    N = 300
    pret, lrtt = (0.6, 0.3)
    e = _setup(lmm, N, pret=pret, lrtt=lrtt)
    rng = np.random.default_rng(9)
    lin = (rng.integers(0, nlin + 1, N)[:, None] == np.arange(1, nlin + 1)[None, :]).astype(float)      # cluster indicators, one class dropped
    cov = rng.standard_normal((N, 2))
    e.lineage_setup(lin, cov)
    labels = ["L%d" % i for i in range(nlin)]
    sizes = [1500, 1, 1030, 2047]
    blocks = [_rows(N, v, 300 + i) for i, v in enumerate(sizes)]
    want = []
    for bits, counts, blob, off in blocks:
        r = e.lmm_batch(bits) if lmm else e.glm_batch(bits)
        if lmm:
            r = mask_like_fit_lmm(r)
        afs = counts.astype(np.float64) / N
        on = (afs >= 0.01) & (afs <= 0.99)
        keys = ("prep", "pvalue", "beta", "bse", "frac_h2") if lmm else ("prep", "pvalue", "kbeta", "bse", "intercept")
        cols = [afs] + [np.where(on, r[k], np.nan) for k in keys]
        flags = np.where(on, r["flags"], np.uint32(1 | (1 << 16))).astype(np.uint32)
        betas = valid = None
        if not lmm:
            betas = r["betas"]; valid = (on & (np.isfinite(r["kbeta"]) | np.isfinite(r["pvalue"]))).astype(np.uint8)
        pf = (flags & (1 << 16)) != 0; ft = (flags & (1 << 17)) != 0
        lineage = np.full(len(on), -1, np.int32)
        if lmm and not per_variant:
            lineage[~pf & ~ft] = int(e.lineage_batch(bits[-1:])[0])
        else:
            need = on & ~pf & (~ft if lmm else ((flags & (1 << 6)) == 0))
            if need.any():
                lineage[need] = e.lineage_batch(bits[need])
        order = np.arange(len(on))
        if lmm:
            order = np.concatenate([order[pf], order[~pf]])
        show = np.ones(len(on), bool) if print_filtered else (~pf & ~ft)
        sel = order[show[order]]
        want.append(RowFormatter(labels).format(blob, off, sel, cols, flags, betas, valid, lineage) if len(sel) else b"")
    job = Job(e, lmm, print_filtered, lineage_labels=labels, lineage_per_variant=per_variant)
    got = []
    for bits, counts, blob, off in blocks:
        job.submit(bits, counts, blob, off)
        while job.pending() > 2:
            got.append(bytes(job.collect()[0]))
    while job.pending():
        got.append(bytes(job.collect()[0]))
    job.close(); e.close()
    assert got == want
    text = b"".join(got)
    assert sum(text.count(b"\t" + l.encode() + b"\t") for l in labels) > 20, "no lineage label was printed"


@pytest.mark.parametrize("lmm", [True, False])
@pytest.mark.parametrize("N", [300, 303, 311])
def test_job_stream_patterns_and_sample_lists(lmm, N):
    """sh_job_set_patterns / sh_job_set_samples (round 6).  Patterns: base64(md5(int64 presence vector)) + newline of every TESTED row of a
    block, in input order, from the device's md5 (csrc/job_kernels.hip k_job_md5) against hashlib through pyseer_amd.input.hash_pattern
    (= the reference's, pyseer/input.py:710-723).  N = 300 / 303 / 311: the tail of the message holds 4 / 7 / 7 samples -- with 7 the length
    goes into a block of its own.  Sample lists (utils.py:96-98): carriers, then the others, names sorted, in front of the notes."""
    import hashlib, binascii
    from pyseer_amd.engine import Engine, Job, pack_variants
    from pyseer_amd.input import hash_pattern
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(N)
    e = Engine(N); e.set_af_filter(0.01, 0.99)
    y = (rng.random(N) < 0.4).astype(float)
    if lmm:
        from pyseer_amd.lmm import initialise_lmm_arrays
        G_ = (rng.random((400, N)) < 0.3).astype(float)
        U, S, h2, nll, Cc = initialise_lmm_arrays(G_.T @ G_, y)
        e.lmm_setup(U, S, y, Cc, h2, filter_pvalue=0.7, lrt_pvalue=0.5)
    else:
        W = rng.standard_normal((N, 2))
        e0 = np.zeros((0, 0))
        e.glm_setup(y, W, False, fit_null(y, W, e0, False).llf, fit_null(y, W, e0, False, firth=True), 0.7, 0.5)
    names = ["s%03d" % int(x) for x in rng.permutation(N)]                       # (not in sorted order)
    order = sorted(range(N), key=lambda i: names[i])
    sizes = [1300, 1, 2050]
    job = Job(e, lmm, False, patterns=True, sample_names=names)
    got_text, got_pat, blocks = [], [], []
    for i, v in enumerate(sizes):
        b = _rows(N, v, 500 + i); blocks.append(b)
        job.submit(*b)
    while job.pending():
        t, c, _ = job.collect(); got_text.append(bytes(t)); got_pat.append((bytes(job.patterns()), c))
    job.close()
    for (bits, counts, blob, off), text, (pat, c) in zip(blocks, got_text, got_pat):
        K = np.unpackbits(bits, axis=1, bitorder="little")[:, :N]
        r = e.lmm_batch(bits) if lmm else e.glm_batch(bits)
        afs = counts / N
        pf = ~((afs >= 0.01) & (afs <= 0.99)) | ((r["flags"] & (1 << 16)) != 0)
        want = b"".join(hash_pattern(K[v].astype(np.int64)) for v in np.flatnonzero(~pf))
        assert c[1] == int((~pf).sum()) and len(pat) == 25 * c[1]
        assert pat == want
        for line in text.splitlines():
            f = line.split(b"\t")
            v = int(f[0][1:7])
            ks = ",".join(names[i] for i in order if K[v, i]); nks = ",".join(names[i] for i in order if not K[v, i])
            assert f[-3].decode() == ks and f[-2].decode() == nks, (v, f[-3][:60], ks[:60])
    e.close()
    assert sum(len(t) for t in got_text) > 0


@pytest.mark.parametrize("lmm", [True, False])
@in_a_process_of_its_own
def test_the_librarys_own_block_loop_over_a_packed_cache(lmm, tmp_path):
    """sh_job_run_packed (round 6, csrc/job_run.inc): the block loop of a `--load-packed` stream inside the library -- index pass, merging of
    stored blocks to block_rows, ranges of the rows for `part_n` devices, DMA windows of the mapping (registered page-disjoint, let go by a
    janitor thread), submit / collect, writes to descriptors.  Against the same blocks pushed through Job.submit / collect from Python:
    same text, same pattern text, same counters; the parts of a three-way split concatenate to the single stream; a zero-carrier row is
    announced on stderr as read_variant does (pyseer/input.py:441-443); a truncated file is refused with the iterator's message."""
    from pyseer_amd.engine import Job
    from pyseer_amd.input import PackedCacheWriter
    from pyseer_amd import _abi
    N = 300
    e = _setup(lmm, N, pret=0.6, lrtt=0.4)
    names = ["s%d" % i for i in range(N)]
    sizes = [700, 40, 2100, 1, 300, 5000, 64, 1200]                  # small stored blocks: several share a page of the mapping
    blocks = [_rows(N, v, 700 + i) for i, v in enumerate(sizes)]
    path = str(tmp_path / "k.seerpack")
    w = PackedCacheWriter(path, names)
    for bits, counts, blob, off in blocks:
        w.write_block(blob, off, counts, bits)
    w.close()

    def by_python(block_rows):
        job = Job(e, lmm, False, patterns=True)
        text, pat, cnt = [], [], [0, 0, 0]
        acc, rows = [], 0
        groups = []
        for b in blocks:
            acc.append(b); rows += b[1].shape[0]
            if rows >= block_rows:
                groups.append(acc); acc, rows = [], 0
        if acc:
            groups.append(acc)
        for g in groups:
            bits = np.concatenate([x[0] for x in g]); counts = np.concatenate([x[1] for x in g]); blob = b"".join(bytes(x[2]) for x in g)
            base = np.cumsum([0] + [len(x[2]) for x in g[:-1]])
            off = np.concatenate([x[3][:-1] + bb for x, bb in zip(g, base)] + [np.array([len(blob)], dtype=np.int64)])
            job.submit(bits, counts, blob, off)
            t, c, _ = job.collect(); text.append(bytes(t)); pat.append(bytes(job.patterns()))
            for a in range(3):
                cnt[a] += c[a]
        job.close()
        return b"".join(text), b"".join(pat), tuple(cnt), len(groups)

    def by_library(block_rows, part=(0, 1), dma=True):
        job = Job(e, lmm, False, patterns=True)
        fo, fp = str(tmp_path / "o.tsv"), str(tmp_path / "p.txt")
        with open(fo, "wb") as a, open(fp, "wb") as b:
            c = job.run_packed(path, part, block_rows, use_dma=dma, out_fd=a.fileno(), pat_fd=b.fileno())
        job.close()
        return open(fo, "rb").read(), open(fp, "rb").read(), c[:3], c[3]

    for block_rows in (1, 1000, 4000):
        want = by_python(block_rows)
        for dma in (True, False):
            got = by_library(block_rows, dma=dma)
            assert got == want, (block_rows, dma, got[2:], want[2:])
        parts = [by_library(block_rows, (i, 3)) for i in range(3)]
        assert b"".join(x[0] for x in parts) == want[0] and b"".join(x[1] for x in parts) == want[1]
        assert tuple(sum(x[2][a] for x in parts) for a in range(3)) == want[2] and sum(x[3] for x in parts) == want[3]
    assert len(want[0]) > 1000 and len(want[1]) == 25 * want[2][1]
    # several DMA windows in this small cache (route key dma_window: bytes registered at a time): merged groups whose stored blocks lie in two
    # windows are copied range by range from both, a block that starts on the page a window ends on belongs to none and its group is staged
    # (one window size: every call registers and lets go a dozen small ranges of the same re-mapped file -- enough of that for one session)
    for window in (30000,):
        os.environ["SEERHIP_ROUTE"] = "dma_window=%d" % window
        try:
            for block_rows in (1000, 4000):
                assert by_library(block_rows) == by_python(block_rows), (window, block_rows)
            parts = [by_library(4000, (i, 2)) for i in range(2)]
        finally:
            del os.environ["SEERHIP_ROUTE"]
        assert b"".join(x[0] for x in parts) == want[0] and b"".join(x[1] for x in parts) == want[1]
    # a truncated cache: refused, with the iterator's message
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:len(raw) - 5000])
    job = Job(e, lmm, False)
    with pytest.raises(_abi.SeerHipError) as ei:
        job.run_packed(path, (0, 1), 1000, out_fd=os.open(os.devnull, os.O_WRONLY))
    assert "truncated packed cache" in str(ei.value)
    job.close(); e.close()


@pytest.mark.parametrize("lmm", [True, False])
@in_a_process_of_its_own
def test_the_block_loop_writes_large_texts_through_its_writer_thread(lmm, tmp_path):
    """Every row printed (the reference's default thresholds; here --print-filtered too): a block's text is several MB, formatted in parts and
    gathered, and sh_job_run_packed hands the gather buffer itself to its writer thread (csrc/job_run.inc OutWriter: detached, a finished one
    attached in its place) -- seven blocks, so buffers are recycled, between small ones that travel as copies.  Same bytes, same order as the
    blocks collected one by one from Python; pattern text too."""
    from pyseer_amd.engine import Job
    from pyseer_amd.input import PackedCacheWriter
    N = 300
    e = _setup(lmm, N, pret=1.0, lrtt=1.0)
    names = ["s%d" % i for i in range(N)]
    sizes = [30000, 50, 30000, 30000, 7, 30000, 30000, 30000, 30000, 900]
    blocks = [_rows(N, v, 1700 + i) for i, v in enumerate(sizes)]
    path = str(tmp_path / "big.seerpack")
    w = PackedCacheWriter(path, names)
    for bits, counts, blob, off in blocks:
        w.write_block(blob, off, counts, bits)
    w.close()
    job = Job(e, lmm, True, patterns=True)
    text, pat, cnt = [], [], [0, 0, 0]
    for bits, counts, blob, off in blocks:
        job.submit(bits, counts, blob, off)
        t, c, _ = job.collect(); text.append(bytes(t)); pat.append(bytes(job.patterns()))
        for a in range(3):
            cnt[a] += c[a]
    job.close()
    want = b"".join(text), b"".join(pat)
    assert max(len(t) for t in text) > (1 << 21) and cnt[2] == sum(sizes)          # every row printed; the large blocks' texts are over 2 MB
    job = Job(e, lmm, True, patterns=True)
    fo, fp = str(tmp_path / "o.tsv"), str(tmp_path / "p.txt")
    with open(fo, "wb") as a, open(fp, "wb") as b:
        c = job.run_packed(path, (0, 1), 1, out_fd=a.fileno(), pat_fd=b.fileno())
    job.close()
    assert (open(fo, "rb").read(), open(fp, "rb").read()) == want and tuple(c[:3]) == tuple(cnt) and c[3] == len(sizes)
    # a descriptor that cannot be written to: the stream ends with an error, not with a short file
    job = Job(e, lmm, True)
    rd = os.open(fo, os.O_RDONLY)
    from pyseer_amd import _abi
    with pytest.raises(_abi.SeerHipError) as ei:
        job.run_packed(path, (0, 1), 1, out_fd=rd)
    os.close(rd)
    assert "writing the output failed" in str(ei.value)
    job.close(); e.close()


@in_a_process_of_its_own
def test_the_block_loop_on_edge_caches(tmp_path):
    """sh_job_run_packed on the edges: a cache without a single block, one stored block shared out to three parts (two of them own nothing),
    and more parts than rows -- counters zero where nothing is owned, the parts still concatenate to the whole."""
    from pyseer_amd.engine import Job
    from pyseer_amd.input import PackedCacheWriter
    N = 300
    e = _setup(True, N, pret=1.0, lrtt=1.0)
    names = ["s%d" % i for i in range(N)]

    def run(path, part):
        job = Job(e, True, True)
        fo = str(tmp_path / "o.tsv")
        with open(fo, "wb") as a:
            c = job.run_packed(path, part, 1000, out_fd=a.fileno())
        job.close()
        return open(fo, "rb").read(), c
    empty = str(tmp_path / "empty.seerpack")
    PackedCacheWriter(empty, names).close()
    assert run(empty, (0, 1)) == (b"", (0, 0, 0, 0)) and run(empty, (2, 3)) == (b"", (0, 0, 0, 0))
    one = str(tmp_path / "one.seerpack")
    w = PackedCacheWriter(one, names)
    bits, counts, blob, off = _rows(N, 5, 901)
    w.write_block(blob, off, counts, bits); w.close()
    whole = run(one, (0, 1))
    assert whole[1][3] == 1 and whole[1][0] + whole[1][1] == 5 and whole[0].count(b"\n") == 5
    for n in (3, 8):
        parts = [run(one, (i, n)) for i in range(n)]
        assert b"".join(p[0] for p in parts) == whole[0]
        assert sum(p[1][3] for p in parts) == 1 and sum(1 for p in parts if p[1][3] == 0) == n - 1
    e.close()
