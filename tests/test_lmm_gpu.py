"""GPU parity of the LMM path (HIP kernels through the C ABI) against the reference's golden vectors and the CPU oracle.
Tolerance: 1e-6 relative on beta / bse / frac_h2 / p-values (BASELINE.json north_star); flags bit-exact."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-6


def close(a, b, rtol=RTOL, atol=0.0, what=""):
    a = np.atleast_1d(np.asarray(a, dtype=float)); b = np.atleast_1d(np.asarray(b, dtype=float))
    assert a.shape == b.shape
    with np.errstate(invalid="ignore"):
        ok = (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))) | \
             (np.abs(a - b) <= atol + rtol * np.abs(b))
    assert ok.all(), "%s mismatch at %s: got %s want %s" % (what, np.where(~ok)[0][:5], a[~ok][:5], b[~ok][:5])


@pytest.fixture(scope="module")
def engine_mod():
    from pyseer_amd.engine import Engine, pack_variants
    return Engine, pack_variants


def _perfect_fit(Kv, y):
    return (Kv == y).all(axis=1) | (Kv == 1 - y).all(axis=1)


def test_lmm_reference_unit_pins(engine_mod):
    Engine, pack = engine_mod
    d = np.load(os.path.join(G, "lmm_unit.npz"))
    for tag in ("nocov", "cov"):
        e = Engine(50)
        e.lmm_setup(d[tag + "_U"], d[tag + "_S"], d[tag + "_y"], d[tag + "_covar"], float(d[tag + "_h2"]))
        for kk, bb in ((tag + "_k", tag + "_blk"), (tag + "_badk", tag + "_blk_bad")):
            r = e.lmm_batch(pack(d[kk].reshape(1, -1)))
            close([r["beta"][0], r["bse"][0], r["frac_h2"][0], r["pvalue"][0]], d[bb][0], what=tag + kk)
        r = e.lmm_batch(pack(d[tag + "_k"].reshape(1, -1)))
        close(r["prep"], [d[tag + "_prep_binary"]])
        with pytest.raises(KeyError):
            e.lmm_setup(d[tag + "_U"], d[tag + "_S"], d[tag + "_y"], d[tag + "_covar"], 1.0)
        with pytest.raises(AssertionError):
            e.lmm_batch(np.zeros((1, 1), dtype=np.uint8))
        e.close()
    # the literals of the reference's tests/lmm_test.py:407-414, through the HIP path
    e = Engine(50)
    e.lmm_setup(d["nocov_U"], d["nocov_S"], d["nocov_y"], d["nocov_covar"], float(d["nocov_h2"]))
    r = e.lmm_batch(pack(d["nocov_k"].reshape(1, -1)))
    assert abs(r["beta"][0] - 0.15136876) < 1e-7 and abs(r["bse"][0] - 0.14208536) < 1e-7
    assert abs(r["frac_h2"][0] - 0.15198184) < 1e-7 and abs(r["pvalue"][0] - 0.29205322) < 1e-7


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "lmm_N*.npz"))))
@pytest.mark.parametrize("limbs", [0, 4, 5, 6])          # 0 = the automatic count (what ships and what the bench times); 4 = its usual choice
def test_lmm_golden_blocks(engine_mod, path, limbs):
    Engine, pack = engine_mod
    d = np.load(path)
    Kv = d["Kv"]; N = int(d["N"]); y = d["y"]
    bits = pack(Kv)
    noise = _perfect_fit(Kv.astype(float), y) if not int(d["continuous"]) else np.zeros(Kv.shape[0], bool)
    e = Engine(N)
    for key in d["h2_keys"]:
        h2 = float(str(key).split("_")[1])
        if abs(h2 - float(d["h2"])) < 1e-6:
            h2 = float(d["h2"])
        e.lmm_setup(d["U"], d["S"], y, d["covar"], h2, continuous=bool(d["continuous"]), n_limbs=limbs)
        r = e.lmm_batch(bits)
        want = d["blk_" + str(key)]
        close(r["beta"], want[:, 0], atol=1e-12, what="beta")
        close(r["frac_h2"], want[:, 2], atol=1e-9, what="frac_h2")
        ok = ~noise & ~np.isnan(want[:, 1])
        close(r["bse"][ok], want[ok, 1], what="bse"); close(r["pvalue"][ok], want[ok, 3], atol=1e-300, what="p")
    if "mis_U" in d.files:          # --load-lmm cache built with other covariates (run_test.sh:47)
        e.lmm_setup(d["mis_U"], d["mis_S"], y, d["covar"], float(d["mis_h2"]), n_limbs=limbs)
        r = e.lmm_batch(bits)
        want = d["mis_blk"]; ok = ~noise & ~np.isnan(want[:, 1])
        close(r["beta"], want[:, 0], atol=1e-12); close(r["frac_h2"], want[:, 2], atol=1e-9)
        close(r["bse"][ok], want[ok, 1]); close(r["pvalue"][ok], want[ok, 3], atol=1e-300)
    e.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "lmm_N*.npz"))))
def test_lmm_golden_orchestration(engine_mod, path):
    """fit_lmm (lmm.py:125-226): AF filter, prefilter (>=), LRT filter (>=), notes, prefilter/filter booleans."""
    Engine, pack = engine_mod
    from pyseer_amd.lmm import mask_like_fit_lmm
    d = np.load(path)
    Kv = d["Kv"]; N = int(d["N"]); y = d["y"]
    noise = _perfect_fit(Kv.astype(float), y) if not int(d["continuous"]) else np.zeros(Kv.shape[0], bool)
    e = Engine(N)
    e.set_af_filter(0.01, 0.99)
    for o, (fp, lp) in (("o1", (1.0, 1.0)), ("o2", (0.05, 0.01))):
        e.lmm_setup(d["U"], d["S"], y, d["covar"], float(d["h2"]), continuous=bool(d["continuous"]), filter_pvalue=fp,
                    lrt_pvalue=lp)
        r = mask_like_fit_lmm(e.lmm_batch(pack(Kv)))
        order = np.array([int(s[1:]) for s in d[o + "_order"]])
        rows = d[o + "_rows"]; nz = ~noise[order]
        got = np.stack([r["prep"], r["pvalue"], r["beta"], r["bse"], r["frac_h2"]], axis=1)[order]
        close(got[nz], rows[nz], atol=1e-300, what=o)
        fl = r["flags"][order]
        assert ((fl & 0x1FF)[nz] == d[o + "_notes"][nz]).all()
        assert (((fl >> 16) & 1) == d[o + "_prefilter"]).all()
        assert (((fl >> 17) & 1)[nz] == d[o + "_filter"][nz]).all()
    e.close()


def _random_lmm(N, D, seed, V):
    rng = np.random.default_rng(seed)
    k = N - D
    U = rng.standard_normal((N, k)) / np.sqrt(N)          # the algebra does not need orthonormal U
    S = np.sort(rng.gamma(0.5, 2.0, k))[::-1].copy()
    covar = np.ones((N, 1)) if D == 1 else np.c_[rng.standard_normal((N, D - 1)), np.ones((N, 1))]
    y = (rng.random(N) < 0.4).astype(float)
    af = rng.uniform(0.02, 0.98, V)
    Kv = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    return U, S, covar, y, Kv


@pytest.mark.parametrize("N,D,V", [(257, 1, 300), (1000, 3, 520), (777, 2, 64)])
def test_lmm_vs_oracle_random(engine_mod, N, D, V):
    Engine, pack = engine_mod
    from oracle import oracle as orc
    U, S, covar, y, Kv = _random_lmm(N, D, 1234 + N, V)
    L = orc.LmmOracle(U, S, y, covar)
    wb, ws, wf, wp = L.block(0.37, Kv.astype(float))
    e = Engine(N)
    e.lmm_setup(U, S, y, covar, 0.37)
    r = e.lmm_batch(pack(Kv))
    close(r["beta"], wb, atol=1e-12, what="beta"); close(r["bse"], ws, what="bse")
    close(r["frac_h2"], wf, atol=1e-9, what="frac"); close(r["pvalue"], wp, atol=1e-300, what="p")
    for v in range(0, V, 37):
        pr, bad = orc.pre_filtering(y, Kv[v].astype(float), False)
        close(r["prep"][v], pr, what="prep"); assert bool(r["flags"][v] & 4) == bad
    e.close()


def test_lmm_full_size_properties(engine_mod):
    """BASELINE config size (N=5000): oracle on a sample + size-independent properties on the whole batch."""
    Engine, pack = engine_mod
    import torch
    from oracle import oracle as orc
    N, D, V = 5000, 1, 2048
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    U = (torch.randn((N, N - 1), generator=g, device="cuda", dtype=torch.float64) / np.sqrt(N)).cpu().numpy()
    rng = np.random.default_rng(9)
    S = np.sort(rng.gamma(0.5, 2.0, N - 1))[::-1].copy()
    y = (rng.random(N) < 0.4).astype(float); covar = np.ones((N, 1))
    af = rng.uniform(0.02, 0.98, V)
    Kv = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    Kv[1] = 1 - Kv[0]                       # complement: beta flips sign, bse / p identical (U~^T 1 = 0)
    Kv[2] = Kv[0]                           # duplicate: identical outputs
    e = Engine(N)
    e.lmm_setup(U, S, y, covar, 0.3)
    r5 = e.lmm_batch(pack(Kv))
    close(r5["beta"][1], -r5["beta"][0], rtol=1e-9); close(r5["bse"][1], r5["bse"][0], rtol=1e-9)
    close(r5["pvalue"][1], r5["pvalue"][0], rtol=1e-8)
    for f in ("beta", "bse", "pvalue", "frac_h2", "prep"):
        assert r5[f][2] == r5[f][0]
    e.lmm_setup(U, S, y, covar, 0.3, n_limbs=6)
    r6 = e.lmm_batch(pack(Kv))
    close(r5["beta"], r6["beta"], rtol=1e-9); close(r5["bse"], r6["bse"], rtol=1e-9)
    close(r5["pvalue"], r6["pvalue"], rtol=1e-7)
    L = orc.LmmOracle(U, S, y, covar)
    idx = np.arange(0, V, 64)
    wb, ws, wf, wp = L.block(0.3, Kv[idx].astype(float))
    close(r5["beta"][idx], wb, atol=1e-12); close(r5["bse"][idx], ws); close(r5["frac_h2"][idx], wf, atol=1e-9)
    close(r5["pvalue"][idx], wp, atol=1e-300)
    # device-resident entry point gives the same numbers
    e.use_torch_stream()
    bt = torch.from_numpy(pack(Kv)).cuda()
    out, fl = e.lmm_batch_dev(bt)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy()[2], r6["beta"]) and np.array_equal(fl.cpu().numpy().astype(np.uint32), r6["flags"])
    e.close()


def test_sharded_engine_matches_single(engine_mod):
    """Multi-GPU sharding path (one context + host thread per device; here two contexts on device 0): same numbers, input order."""
    Engine, pack = engine_mod
    from pyseer_amd.parallel import ShardedEngine
    d = np.load(os.path.join(G, "lmm_N300_D3.npz"))
    bits = pack(np.tile(d["Kv"], (5, 1))[:301])
    e = Engine(300); e.lmm_setup(d["U"], d["S"], d["y"], d["covar"], float(d["h2"]))
    want = e.lmm_batch(bits); e.close()
    s = ShardedEngine(300, [0, 0, 0])
    s.lmm_setup(d["U"], d["S"], d["y"], d["covar"], float(d["h2"]))
    got = s.lmm_batch(bits); s.close()
    for k in want:
        assert np.array_equal(got[k], want[k], equal_nan=True), k


@pytest.mark.parametrize("N,D,cont", [(129, 6, False), (640, 9, True), (64, 1, True)])
def test_lmm_many_covariates_and_continuous_vs_oracle(engine_mod, N, D, cont):
    """Wider covariate panels (DP = 8, 16 residual kernels), continuous-phenotype prefilter, sample counts at tile edges."""
    Engine, pack = engine_mod
    from oracle import oracle as orc
    U, S, covar, y, Kv = _random_lmm(N, D, 99 + N, 200)
    if cont:
        y = np.random.default_rng(N).standard_normal(N)
    L = orc.LmmOracle(U, S, y, covar)
    wb, ws, wf, wp = L.block(0.6, Kv.astype(float))
    e = Engine(N)
    e.lmm_setup(U, S, y, covar, 0.6, continuous=cont)
    r = e.lmm_batch(pack(Kv))
    close(r["beta"], wb, atol=1e-12); close(r["bse"], ws); close(r["frac_h2"], wf, atol=1e-9); close(r["pvalue"], wp, atol=1e-300)
    for v in range(0, 200, 17):
        pr, bad = orc.pre_filtering(y, Kv[v].astype(float), cont)
        close(r["prep"][v], pr, what="prep"); assert bool(r["flags"][v] & 4) == bad
    assert e.lmm_batch(np.zeros((0, (N + 63) // 64 * 8), dtype=np.uint8))["beta"].shape == (0,)
    e.close()


def test_pattern_dedup_is_exact_and_transparent(engine_mod):
    """sh_set_dedup: duplicated patterns are tested once, outputs identical to the plain path (LMM and fixed effects)."""
    Engine, pack = engine_mod
    d = np.load(os.path.join(G, "lmm_N300_D3.npz"))
    rng = np.random.default_rng(3)
    Kv = d["Kv"][rng.integers(0, d["Kv"].shape[0], 1500)]           # 64 distinct patterns, 1500 rows, shuffled
    bits = pack(Kv)
    e = Engine(300); e.lmm_setup(d["U"], d["S"], d["y"], d["covar"], float(d["h2"]), filter_pvalue=0.5, lrt_pvalue=0.5)
    want = e.lmm_batch(bits)
    e.set_dedup(True)
    got = e.lmm_batch(bits)
    assert e.dedup_info() == len(np.unique(bits, axis=0))
    for k in want:
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    e.close()
    g = np.load(os.path.join(G, "glm_N300_q10.npz"))
    Kg = g["K"][rng.integers(0, g["K"].shape[0], 700)]
    e = Engine(300); e.set_af_filter(0.01, 0.99)
    e.glm_setup(g["y"], g["m"], False, float(g["null_llf"]), float(g["null_firth"]))
    want = e.glm_batch(pack(Kg))
    e.set_dedup(True)
    got = e.glm_batch(pack(Kg))
    assert e.dedup_info() == len(np.unique(pack(Kg), axis=0))
    for k in want:
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    e.close()


def test_long_rows_use_the_gather_repack(engine_mod):
    """More than 8192 samples per row: the repack falls back to the per-thread gather; results must not depend on the path.
    Checked on the similarity accumulator (exact integers; no O(N^3) setup needed at this N)."""
    from pyseer_amd.engine import Engine, pack_variants
    rng = np.random.default_rng(5)
    n, V = 8300, 700
    Kv = (rng.random((V, n)) < 0.3).astype(np.uint8)
    e = Engine(n)
    e.sim_begin(); e.sim_accumulate(pack_variants(Kv))
    K = e.sim_finish()
    e.close()
    G = Kv.astype(np.float32)
    assert np.array_equal(K, (G.T @ G).astype(np.float64))


@pytest.mark.parametrize("D", [3, 6])
def test_lmm_variant_in_the_covariate_span_is_zeroed(engine_mod, D):
    """rotate() zeroes a column that the covariates explain (std <= 1e-10 after regressing them out, lmm_cov.py:179-181): a k-mer
    equal to a binary covariate, or to its complement.  The table kernel sees m - sum c^2 cancel and must take the exact path."""
    Engine, pack = engine_mod
    from oracle import oracle as orc
    N = 200
    U, S, covar, y, Kv = _random_lmm(N, D, 7 + D, 40)
    covar = covar.copy()
    covar[:, 0] = (np.random.default_rng(D).random(N) < 0.35).astype(float)       # a binary covariate (the intercept is last)
    Kv = Kv.copy()
    Kv[5] = covar[:, 0].astype(np.uint8)
    Kv[11] = 1 - covar[:, 0].astype(np.uint8)
    L = orc.LmmOracle(U, S, y, covar)
    wb, ws, wf, wp = L.block(0.4, Kv.astype(float))
    e = Engine(N)
    e.lmm_setup(U, S, y, covar, 0.4)
    r = e.lmm_batch(pack(Kv))
    e.close()
    close(r["beta"], wb, atol=1e-12, what="beta"); close(r["bse"], ws, what="bse")
    close(r["frac_h2"], wf, atol=1e-9, what="frac_h2"); close(r["pvalue"], wp, atol=1e-300, what="pvalue")
    # whatever the reference makes of a zeroed column, both sides make the same of it
    assert np.array_equal(np.isnan(r["bse"][[5, 11]]), np.isnan(ws[[5, 11]]))


def test_host_batches_longer_than_one_pipeline_chunk(engine_mod):
    """sh_lmm_batch / sh_glm_batch cut host batches into 2^18-variant chunks on two staging sets; several chunks plus a ragged
    tail must give exactly what single small calls give."""
    Engine, pack = engine_mod
    N, D = 130, 1
    V = (1 << 18) * 2 + 1234
    U, S, covar, y, _ = _random_lmm(N, D, 5, 8)
    rng = np.random.default_rng(11)
    bits = rng.integers(0, 256, (V, (N + 63) // 64 * 8), dtype=np.uint8)
    bits[:, N // 8] &= (1 << (N % 8)) - 1
    bits[:, N // 8 + 1:] = 0
    e = Engine(N)
    e.lmm_setup(U, S, y, covar, 0.3)
    big = e.lmm_batch(bits)
    for lo in (0, (1 << 18) - 7, (1 << 19) - 3, V - 500):
        part = e.lmm_batch(bits[lo:lo + 500])
        for f in ("prep", "pvalue", "beta", "bse", "frac_h2", "flags"):
            assert np.array_equal(big[f][lo:lo + 500], part[f], equal_nan=True), (f, lo)
    W = rng.standard_normal((N, 2))
    e.glm_setup(y, W, False, -80.0, -75.0)
    bigg = e.glm_batch(bits)
    for lo in (0, (1 << 18) - 7, V - 500):
        part = e.glm_batch(bits[lo:lo + 500])
        for f in ("prep", "pvalue", "kbeta", "bse", "intercept", "betas", "flags"):
            assert np.array_equal(bigg[f][lo:lo + 500], part[f], equal_nan=True), (f, lo)
    e.close()


def test_pipelined_host_batches_equal_the_synchronous_ones(engine_mod):
    """sh_lmm_batch_async / sh_glm_batch_async (include/seerhip.h): a call returns with its last chunk still on the device; its arrays are
    complete after the next call on the context (pipelined or not) or after sh_wait.  Blocks of several chunk counts in a row, an empty
    block in between, a synchronous call and a new set-up behind a pending batch: every result bit-identical to the synchronous call's."""
    Engine, pack = engine_mod
    N, D = 130, 1
    U, S, covar, y, _ = _random_lmm(N, D, 5, 8)
    rng = np.random.default_rng(12)
    rb = (N + 63) // 64 * 8
    sizes = [700, (1 << 17) + 5, 0, (1 << 18) + (1 << 17) + 77, 3, 1 << 17]
    blocks = []
    for V in sizes:
        b = rng.integers(0, 256, (V, rb), dtype=np.uint8)
        b[:, N // 8] &= (1 << (N % 8)) - 1
        b[:, N // 8 + 1:] = 0
        blocks.append(b)
    e = Engine(N)
    e.lmm_setup(U, S, y, covar, 0.3)
    want = [e.lmm_batch(b) for b in blocks]
    got = [e.lmm_batch(b, pipelined=True) for b in blocks]
    e.wait()
    for w, g, V in zip(want, got, sizes):
        for f in ("prep", "pvalue", "beta", "bse", "frac_h2", "flags"):
            assert g[f].shape[0] == V and np.array_equal(w[f], g[f], equal_nan=True), (f, V)
    # completion by the next call alone (no wait): the first block's arrays after the second call has returned
    g0 = e.lmm_batch(blocks[1], pipelined=True)
    g1 = e.lmm_batch(blocks[0])                                   # a synchronous call completes the pending batch as well
    for f in ("prep", "pvalue", "beta", "bse", "frac_h2", "flags"):
        assert np.array_equal(want[1][f], g0[f], equal_nan=True) and np.array_equal(want[0][f], g1[f], equal_nan=True), f
    # a new model behind a pending batch: the pending results are delivered first
    g3 = e.lmm_batch(blocks[3], pipelined=True)
    W = rng.standard_normal((N, 2))
    e.glm_setup(y, W, False, -80.0, -75.0)
    for f in ("prep", "pvalue", "beta", "bse", "frac_h2", "flags"):
        assert np.array_equal(want[3][f], g3[f], equal_nan=True), f
    wantg = [e.glm_batch(b) for b in blocks]
    gotg = [e.glm_batch(b, pipelined=True) for b in blocks]
    e.wait()
    for w, g in zip(wantg, gotg):
        for f in ("prep", "pvalue", "kbeta", "bse", "intercept", "betas", "flags"):
            assert np.array_equal(w[f], g[f], equal_nan=True), f
    # rows announced one batch ahead (sh_prefetch_rows: their first chunk is uploaded by the batch before); an announcement that the next
    # batch does not honour (other rows; the announced array overwritten in place after its upload) must change nothing
    nz = [b for b in blocks if b.shape[0]]
    gotp = []
    for i, b in enumerate(nz):
        if i + 1 < len(nz):
            e.prefetch(nz[i + 1])
        gotp.append(e.glm_batch(b, pipelined=True))
    e.wait()
    for w, g in zip([w for w, b in zip(wantg, blocks) if b.shape[0]], gotp):
        for f in ("prep", "pvalue", "kbeta", "bse", "intercept", "betas", "flags"):
            assert np.array_equal(w[f], g[f], equal_nan=True), ("announced", f)
    e.prefetch(nz[1])
    a = e.glm_batch(nz[0], pipelined=True)
    c = e.glm_batch(nz[3], pipelined=True)                         # not the announced rows
    e.wait()
    scratch = nz[1].copy()
    e.prefetch(scratch)
    e.glm_batch(nz[0], pipelined=True)
    scratch[:] = nz[4][:1].repeat(scratch.shape[0], axis=0) if nz[4].shape[0] < scratch.shape[0] else nz[4][:scratch.shape[0]]
    d = e.glm_batch(scratch, pipelined=True)                      # same pointer, other rows than were uploaded
    e.wait()
    wd = e.glm_batch(scratch)
    wn = [w for w, b in zip(wantg, blocks) if b.shape[0]]
    for f in ("prep", "pvalue", "kbeta", "bse", "intercept", "betas", "flags"):
        assert np.array_equal(wn[0][f], a[f], equal_nan=True) and np.array_equal(wn[3][f], c[f], equal_nan=True), ("other rows", f)
        assert np.array_equal(wd[f], d[f], equal_nan=True), ("overwritten", f)
    e.close()


def test_fit_lmm_block_drop_in_reads_like_the_reference_test(engine_mod):
    """tests/lmm_test.py:395-420 (TestFitLmmBlock) with pyseer_amd.lmm.fit_lmm_block in place of pyseer.lmm.fit_lmm_block."""
    from pyseer_amd.lmm import fit_lmm_block, LmmState
    d = np.load(os.path.join(G, "lmm_unit.npz"))
    y = LmmState(d["nocov_U"], d["nocov_S"], d["nocov_y"], d["nocov_covar"])
    z = float(d["nocov_h2"])
    variant_mat = d["nocov_k"].astype(float).reshape(-1, 1)
    result = fit_lmm_block(y, z, variant_mat)
    assert abs(result['beta'][0] - 0.15136876) < 5e-8
    assert abs(result['bse'][0] - 0.14208536) < 5e-8
    assert abs(result['frac_h2'][0] - 0.15198184) < 5e-8
    assert abs(result['p_values'][0] - 0.29205322) < 5e-8
    with pytest.raises(KeyError):                       # impossibly high h2
        fit_lmm_block(y, 1, variant_mat)
    with pytest.raises(AssertionError):                 # shape mismatch
        fit_lmm_block(y, z, variant_mat[:10])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "lmm_N*.npz"))))
def test_fit_lmm_drop_in_matches_golden_orchestration(engine_mod, path):
    """pyseer_amd.lmm.fit_lmm with the reference's signature: tuple stream in, LMM tuples out, filtered variants first
    (goldens: fit_lmm itself, tests/golden/make_golden.py run_lmm_case orchestrate)."""
    from pyseer_amd.lmm import fit_lmm, LmmState
    from pyseer_amd.classes import LMM, NOTE_ORDER
    d = np.load(path)
    Kv = d["Kv"].astype(float); y = d["y"]; cont = bool(d["continuous"])
    noise = _perfect_fit(Kv, y) if not cont else np.zeros(Kv.shape[0], bool)
    st = LmmState(d["U"], d["S"], y, d["covar"])
    af = Kv.mean(axis=1)
    keep = (af >= 0.01) & (af <= 0.99)
    for o, (fp, lp) in (("o1", (1.0, 1.0)), ("o2", (0.05, 0.01))):
        variants = [(LMM("v%d" % i, b"x" if keep[i] else None, af[i], np.nan, np.nan, np.nan, np.nan, np.nan, np.nan, [], [],
                         set(), True, True), y, Kv[i] if keep[i] else None) for i in range(Kv.shape[0])]
        mat = np.where(keep[None, :], Kv.T, 0.0).copy()
        got = fit_lmm(st, float(d["h2"]), variants, mat, False, None, d["covar"], cont, fp, lp)
        order = [str(s) for s in d[o + "_order"]]
        assert [g.kmer for g in got] == order
        rows = d[o + "_rows"]
        for j, g in enumerate(got):
            i = int(g.kmer[1:])
            assert g.prefilter == bool(d[o + "_prefilter"][j])
            if noise[i]:
                continue
            vals = np.array([g.prep, g.pvalue, g.kbeta, g.bse, g.frac_h2], dtype=float)
            w = rows[j]
            ok = (np.isnan(vals) & np.isnan(w)) | (np.abs(vals - w) <= 1e-6 * np.abs(w) + 1e-300)
            assert ok.all(), (g.kmer, vals, w)
            assert set(n for b, n in enumerate(NOTE_ORDER) if (int(d[o + "_notes"][j]) >> b) & 1) == g.notes
            assert g.filter == bool(d[o + "_filter"][j])
        assert (mat[:, ~keep] == 0).all()


def test_contexts_release_their_memory(engine_mod):
    """sh_destroy gives back every allocation (workspace, staging, tables, Firth state, similarity accumulator)."""
    import torch
    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("free device memory is a device-wide figure: other xdist workers allocate while this test measures")
    Engine, pack = engine_mod
    rng = np.random.default_rng(3)
    N = 300
    U, S, covar, y, Kv = _random_lmm(N, 2, 9, 400)
    W = rng.standard_normal((N, 3))

    def cycle():
        e = Engine(N)
        e.lmm_setup(U, S, y, covar, 0.3)
        e.set_dedup(True)
        e.lmm_batch(pack(Kv))
        e.glm_setup(y, W, False, -150.0, -140.0, force_firth=True)
        e.glm_batch(pack(Kv))
        e.sim_begin(); e.sim_accumulate(pack(Kv)); e.sim_finish()
        e.close()

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < (8 << 20)


@pytest.mark.parametrize("frac_rare", [0.02, 0.4, 1.0])
def test_af_compaction_changes_nothing(engine_mod, frac_rare, monkeypatch):
    """AF-filtered variants never read their quadratic form, so when >= 3 % of a batch is filtered only the kept columns are
    contracted (gather -> k_lmm_quadform_i8 -> scatter).  Bit-identical outputs with the compaction on and off, flags and NaNs for
    the filtered rows included, and the oracle on the kept ones."""
    Engine, pack = engine_mod
    from oracle import oracle as orc
    N, V = 900, 1500
    U, S, covar, y, Kv = _random_lmm(N, 2, 99, V)
    rng = np.random.default_rng(5)
    rare = rng.random(V) < frac_rare
    Kv[rare] = (rng.random((int(rare.sum()), N)) < 0.004).astype(np.uint8)
    if frac_rare == 1.0:
        Kv[:] = 0                                                  # nothing is kept at all
    bits = pack(Kv)
    res = []
    for on in ("2", "0", "1"):                                     # 2 = count every batch, 0 = never compact, 1 = default (adaptive)
        monkeypatch.setenv("SEERHIP_ROUTE", "afcompact=" + on)
        e = Engine(N); e.set_af_filter(0.01, 0.99)
        e.lmm_setup(U, S, y, covar, 0.41)
        res.append(e.lmm_batch(bits)); res.append(e.lmm_batch(bits)); e.close()       # the second call of a context knows the first one's count
    for other in res[1:]:
        for f in ("prep", "pvalue", "beta", "bse", "frac_h2"):
            assert np.array_equal(res[0][f], other[f], equal_nan=True), f
        assert np.array_equal(res[0]["flags"], other["flags"])
    af = Kv.mean(axis=1); kept = (af >= 0.01) & (af <= 0.99)
    assert ((res[0]["flags"][~kept] & 0x10001) == 0x10001).all() and np.isnan(res[0]["pvalue"][~kept]).all()
    if kept.any():
        wb, ws, wf, wp = orc.LmmOracle(U, S, y, covar).block(0.41, Kv[kept].astype(float))
        close(res[0]["beta"][kept], wb, atol=1e-12, what="beta"); close(res[0]["pvalue"][kept], wp, atol=1e-300, what="p")


def test_prefilter_compaction_changes_nothing(engine_mod, monkeypatch):
    """--filter-pvalue with --lmm: a pre-filtered variant is never fitted, so its quadratic form is not computed (k_af_keep drops it along
    with the AF-filtered ones).  Same outputs as the reference's fit_lmm (oracle), and identical with the compaction forced, off, adaptive."""
    Engine, pack = engine_mod
    from oracle import oracle as orc
    from pyseer_amd.lmm import mask_like_fit_lmm
    N, V, pret = 700, 2000, 0.05
    U, S, covar, y, Kv = _random_lmm(N, 2, 123, V)
    rng = np.random.default_rng(8)
    causal = rng.random(V) < 0.04
    Kv[causal] = ((rng.random((int(causal.sum()), N)) < 0.2) | ((y == 1) & (rng.random((int(causal.sum()), N)) < 0.35))).astype(np.uint8)
    Kv[::40] = (rng.random((len(Kv[::40]), N)) < 0.004).astype(np.uint8)
    bits = pack(Kv)
    res = []
    for on in ("2", "0", "1"):
        monkeypatch.setenv("SEERHIP_ROUTE", "afcompact=" + on)
        e = Engine(N); e.set_af_filter(0.01, 0.99)
        e.lmm_setup(U, S, y, covar, 0.33, continuous=False, filter_pvalue=pret, lrt_pvalue=0.5)
        res.append(e.lmm_batch(bits)); res.append(e.lmm_batch(bits)); e.close()
    for other in res[1:]:
        for f in ("prep", "pvalue", "beta", "bse", "frac_h2"):
            assert np.array_equal(res[0][f], other[f], equal_nan=True), f
        assert np.array_equal(res[0]["flags"], other["flags"])
    af = Kv.mean(axis=1)
    afm = (~((af >= 0.01) & (af <= 0.99))).astype(np.uint8)
    want = orc.LmmOracle(U, S, y, covar).fit_lmm(0.33, Kv.astype(float), afm, False, pret, 0.5)
    r = mask_like_fit_lmm(res[0])
    assert want["prefilter"].mean() > 0.5 and (want["prefilter"] == 0).sum() > 30
    for f, g in (("prep", "prep"), ("pvalue", "pvalue"), ("beta", "kbeta"), ("bse", "bse"), ("frac_h2", "frac_h2")):
        close(r[f], want[g], atol=1e-12 if f in ("beta", "frac_h2") else 1e-300, what=f)
    assert ((r["flags"] & 0x1FF) == want["notes"]).all()
    assert (((r["flags"] >> 16) & 1) == want["prefilter"]).all() and (((r["flags"] >> 17) & 1) == want["filter"]).all()


def _dev(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    ok = np.isfinite(a) & np.isfinite(b) & (np.abs(b) > 1e-200)
    return float(np.max(np.abs(a[ok] - b[ok]) / np.abs(b[ok]))) if ok.any() else 0.0


def test_extra_limb_pass_is_as_good_as_more_limbs(engine_mod):
    """The a-posteriori bound and the extra-limb pass (include/seerhip.h sh_lmm_bound): with 4 limbs and the pass switched off the
    statistics carry the 4-limb quantisation (~1e-9); with a tolerance below every variant's bound all of them are contracted again with
    the two extra limbs and land where 6 limbs land; the reported bound always covers the measured deviation of xKx (~ 2 x that of bse)."""
    Engine, pack = engine_mod
    from oracle import oracle as orc
    N, D, V = 1000, 1, 384
    U, S, covar, y, Kv = _random_lmm(N, D, 77, V)
    wb, ws, wf, wp = orc.LmmOracle(U, S, y, covar).block(0.37, Kv.astype(float))
    e = Engine(N)
    e.lmm_setup(U, S, y, covar, 0.37, n_limbs=4)
    e.set_lmm_tol(0.0)                                             # never refine
    r0 = e.lmm_batch(pack(Kv)); i0 = e.lmm_info()
    assert i0["refined_last_batch"] == 0 and i0["extra_limbs"] == 2
    d0 = _dev(r0["bse"], ws)
    assert 1e-12 < d0 < 1e-7, d0                                   # the 4-limb error is visible ...
    assert 2 * d0 <= i0["bound_rel_max_last_batch"] * 1.01         # ... and inside the bound
    e.set_lmm_tol(1e-13)                                           # below every bound: every fitted variant is refined
    r1 = e.lmm_batch(pack(Kv)); i1 = e.lmm_info()
    assert i1["refined_last_batch"] == V, i1
    d1 = max(_dev(r1["bse"], ws), _dev(r1["beta"], wb), _dev(r1["pvalue"], wp))
    assert d1 < 2e-12, d1
    assert i1["bound_rel_max_last_batch"] < i0["bound_rel_max_last_batch"] / 6e4
    # a tolerance in between refines only the variants above it, and those are exactly the ones whose 4-limb bound exceeded it
    tol = float(np.median([i0["bound_rel_max_last_batch"], i0["bound_rel_typical"]]))
    e.set_lmm_tol(tol)
    r2 = e.lmm_batch(pack(Kv)); i2 = e.lmm_info()
    assert 0 < i2["refined_last_batch"] < V, (tol, i2)
    same = r2["bse"] == r0["bse"]; refined = r2["bse"] == r1["bse"]
    assert (same | refined).all() and refined.sum() >= i2["refined_last_batch"]
    e.close()


def test_complemented_rows_change_nothing_but_rounding(engine_mod, monkeypatch):
    """Rows with more than N/2 carriers are stored complemented (k_repack_bits); SEERHIP_ROUTE=complement=0 (read by sh_lmm_setup) stores
    them as given.  Same statistics up to the quantisation of G, counts / prefilter / flags identical, with covariates and a continuous
    phenotype as well."""
    Engine, pack = engine_mod
    for N, D, cont in ((300, 1, False), (515, 3, False), (200, 2, True)):
        U, S, covar, y, Kv = _random_lmm(N, D, 5 + N, 256)
        if cont:
            y = np.random.default_rng(1).standard_normal(N)
        Kv[:64] = (np.random.default_rng(2).random((64, N)) < 0.97).astype(np.uint8)      # majority-carrier rows
        e = Engine(N); e.set_af_filter(0.01, 0.99)
        e.lmm_setup(U, S, y, covar, 0.41, continuous=cont, filter_pvalue=0.7)
        a = e.lmm_batch(pack(Kv))
        monkeypatch.setenv("SEERHIP_ROUTE", "complement=0")
        e.lmm_setup(U, S, y, covar, 0.41, continuous=cont, filter_pvalue=0.7)
        b = e.lmm_batch(pack(Kv))
        monkeypatch.delenv("SEERHIP_ROUTE")
        e.close()
        assert np.array_equal(a["flags"], b["flags"]) and np.array_equal(np.isnan(a["prep"]), np.isnan(b["prep"]))
        assert _dev(a["prep"], b["prep"]) < (1e-12 if cont else 1e-300)    # binary: integer counts, identical; Welch sums: by subtraction
        for f in ("beta", "bse", "pvalue", "frac_h2"):
            assert np.array_equal(np.isnan(a[f]), np.isnan(b[f]))
            assert _dev(a[f], b[f]) < 1e-9, (N, f, _dev(a[f], b[f]))


@pytest.mark.parametrize("N,V,limbs", [(520, 700, 0), (660, 513, 4), (680, 512, 4), (760, 100, 5), (1100, 1536, 5), (2049, 600, 4)])
def test_one_wave_and_two_wave_contractions_agree_bit_for_bit(engine_mod, monkeypatch, N, V, limbs):
    """k_lmm_quadform_i8w (one wavefront per SIMD, the default where its conditions hold: here N >= 512) and k_lmm_quadform_i8 (SEERHIP_ROUTE qf=0, read
    when the context is created; also the fallback for fewer than four row tiles) are the same exact integer contraction with the same fp64
    recombination: every output double must be identical, whatever the shape (V not a multiple of the block, several launches' worth of tiles).
    The sample counts cover the one-wave kernel's treatment of the last 128-row tile (NR = 2 ceil(N / 256) row tiles): all padding and dropped
    (520, 1100, 2049), 20 valid rows = one 32-row sub-tile (660), 40 = two (680), 120 = the full tile (760)."""
    Engine, pack = engine_mod
    U, S, covar, y, Kv = _random_lmm(N, 1, 77 + N, V)
    Kv[: V // 8] = (np.random.default_rng(3).random((V // 8, N)) < 0.9).astype(np.uint8)      # majority-carrier rows (stored complemented)
    bits = pack(Kv)
    out = []
    for qf in ("4", "0"):
        monkeypatch.setenv("SEERHIP_ROUTE", "qf=" + qf)
        e = Engine(N)
        e.lmm_setup(U, S, y, covar, 0.33, n_limbs=limbs)
        out.append(e.lmm_batch(bits)); e.close()
    monkeypatch.delenv("SEERHIP_ROUTE")
    a, b = out
    assert np.array_equal(a["flags"], b["flags"])
    for f in ("prep", "beta", "bse", "pvalue", "frac_h2"):
        assert np.array_equal(a[f].view(np.uint64), b[f].view(np.uint64)), (N, f)


def test_covariates_without_intercept_through_the_abi(engine_mod):
    """pyseer always appends the intercept (lmm.py:95-99); the C ABI does not require it.  Without it U~^T 1 != 0, rows must not be
    complemented, and the statistics still follow the oracle."""
    Engine, pack = engine_mod
    from oracle import oracle as orc
    N, V = 400, 200
    U, S, _, y, Kv = _random_lmm(N, 1, 91, V)
    covar = np.random.default_rng(3).standard_normal((N, 2))       # no constant column
    Kv[:50] = (np.random.default_rng(4).random((50, N)) < 0.9).astype(np.uint8)
    wb, ws, wf, wp = orc.LmmOracle(U, S, y, covar).block(0.3, Kv.astype(float))
    e = Engine(N); e.lmm_setup(U, S, y, covar, 0.3)
    r = e.lmm_batch(pack(Kv)); e.close()
    close(r["beta"], wb, atol=1e-12); close(r["bse"], ws); close(r["pvalue"], wp, atol=1e-300)


def test_more_than_8192_samples_repack_fallback(engine_mod):
    """Rows longer than 1 KB take the gather form of the bit repack (k_repack_bits_gather + k_row_flip): low-rank U keeps the oracle cheap."""
    Engine, pack = engine_mod
    from oracle import oracle as orc
    N, k, V = 8300, 48, 96
    rng = np.random.default_rng(12)
    U = np.linalg.qr(rng.standard_normal((N, k + 1)))[0]
    one = np.ones(N) / np.sqrt(N)
    U = U - np.outer(one, one @ U); U = np.linalg.qr(U)[0][:, :k]  # orthogonal to the intercept, as a fresh decomposition is
    S = np.sort(rng.gamma(0.5, 2.0, k))[::-1].copy()
    y = (rng.random(N) < 0.4).astype(float); covar = np.ones((N, 1))
    Kv = (rng.random((V, N)) < rng.uniform(0.02, 0.98, V)[:, None]).astype(np.uint8)
    wb, ws, wf, wp = orc.LmmOracle(U, S, y, covar).block(0.3, Kv.astype(float))
    e = Engine(N); e.lmm_setup(U, S, y, covar, 0.3)
    r = e.lmm_batch(pack(Kv)); e.close()
    close(r["beta"], wb, atol=1e-12); close(r["bse"], ws); close(r["pvalue"], wp, atol=1e-300)
    for v in range(0, V, 13):
        pr, bad = orc.pre_filtering(y, Kv[v].astype(float), False)
        close(r["prep"][v], pr)


def test_lmm_state_shared_between_contexts(engine_mod):
    """sh_lmm_share: a second context that received the per-run state device-to-device gives bit-identical results (covariates,
    complemented rows and the extra-limb pass included); sharing before set-up is refused."""
    Engine, pack = engine_mod
    from pyseer_amd import _abi
    U, S, covar, y, Kv = _random_lmm(515, 3, 61, 300)
    a = Engine(515); b = Engine(515)
    with pytest.raises(_abi.SeerHipError):
        b.lmm_share_from(a)
    a.set_af_filter(0.01, 0.99); b.set_af_filter(0.01, 0.99)
    a.lmm_setup(U, S, y, covar, 0.41, n_limbs=4); a.set_lmm_tol(1e-10)
    b.lmm_share_from(a)
    ra = a.lmm_batch(pack(Kv)); rb = b.lmm_batch(pack(Kv))
    assert a.lmm_info()["refined_last_batch"] == b.lmm_info()["refined_last_batch"] > 0
    for k in ra:
        assert np.array_equal(ra[k], rb[k], equal_nan=True), k
    a.close(); b.close()


def test_sharded_engine_across_real_devices(engine_mod):
    """ShardedEngine over every visible GPU (skipped on a one-GPU box): same numbers as one context, input order kept."""
    Engine, pack = engine_mod
    import torch
    from pyseer_amd.parallel import ShardedEngine
    nd = torch.cuda.device_count()
    if nd < 2:
        pytest.skip("one GPU visible")
    d = np.load(os.path.join(G, "lmm_N300_D3.npz"))
    bits = pack(np.tile(d["Kv"], (40, 1))[:2001])
    e = Engine(300); e.lmm_setup(d["U"], d["S"], d["y"], d["covar"], float(d["h2"]))
    want = e.lmm_batch(bits); e.close()
    s = ShardedEngine(300, list(range(nd)))
    s.lmm_setup(d["U"], d["S"], d["y"], d["covar"], float(d["h2"]))
    got = s.lmm_batch(bits); s.close()
    for k in want:
        assert np.array_equal(got[k], want[k], equal_nan=True), k


def test_spectral_norm_certificate_cannot_underestimate():
    """sh_lmm_setup bounds the quantisation error's spectral norm by trace(E^64)^(1/64) (repeated squaring on the fp64 matrix pipe).  On
    matrices with known spectra -- including the two cases a power iteration is bad at: two near-equal top eigenvalues, and a top pair of
    opposite sign (the iterate oscillates between the two eigenvectors) -- the certificate is never below the true norm and within the
    factor its derivation gives ((n C_32 / 4^32)^(1/64) at worst); the 48-step power iteration (the round-2 estimate) does fall short."""
    from pyseer_amd.engine import Engine
    rng = np.random.default_rng(7)
    n = 700
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    e = Engine(300)
    worst_power = 1.0
    for name, lam in (("wigner", None),
                      ("near-equal top pair", np.concatenate([[1.0, 0.9995], rng.uniform(-0.6, 0.6, n - 2)])),
                      ("opposite top pair", np.concatenate([[1.0, -1.0], rng.uniform(-0.3, 0.3, n - 2)])),
                      ("slowly decaying", 1.0 / (1.0 + 0.002 * np.arange(n)) * np.where(np.arange(n) % 2, -1.0, 1.0))):
        if lam is None:
            A = rng.uniform(-0.25, 0.25, (n, n)); A = np.tril(A, -1); A = A + A.T        # the shape of a quantisation-error matrix
        else:
            A = (Q * lam) @ Q.T
        A = ((A + A.T) / 2).astype(np.float32)
        true = float(np.abs(np.linalg.eigvalsh(A.astype(np.float64))).max())
        up, pw = e.spectral_bound(A, squarings=5)
        worst_power = min(worst_power, pw / true)
        print("%-20s true %.6g  certificate %.6g (x%.4f)  power iteration %.6g (x%.4f)" % (name, true, up, up / true, pw, pw / true))
        assert up >= true * (1.0 - 1e-9), (name, up, true)
        assert up <= true * float(n) ** (1.0 / 64) * 1.001, (name, up, true)             # trace(A^64) <= n |A|^64
    e.close()
    assert worst_power < 0.999                                                           # what the certificate is for

