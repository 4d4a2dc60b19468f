"""The same call again: same bits.  A variant's result is a function of its row and the run's constants -- the job stream, the lanes and the
`--gpus` parts rest on that -- so repeated calls on one context must agree bit for bit, also while another stream keeps the device busy (a
loaded memory system is where a read that is ordered by nothing but time goes wrong).  Round 6 found one that way: k_firth_fast read the next
ring stage's first record a whole iteration after its copy was issued, but behind no wait -- 5 of 60 repeated calls of the single-precision
pass differed at Q <= 3 (profiles/r06/firth_determinism.txt; the ring is four stages deep since).  The shapes here are the ones that showed it
(few covariates: short iterations), through every first-pass route of the one-pass Firth kernels, and the LMM / logistic paths on a small batch."""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _Busy(object):
    """a second context of the library on the same device, testing batches of its own on a thread of its own while the calls under test run
    (the library's own concurrency, as tests/test_glm_gpu.py::test_contexts_on_one_device_run_concurrently_and_agree_bit_for_bit uses it;
    tools/gpu_firth_determinism.py loads the device with matrix products of another framework instead)"""
    def __enter__(self):
        import bench
        from pyseer_amd.engine import Engine, pack_variants
        self.stop = False
        N, q = 1000, 10
        y, W, nl, nf = bench.synth_glm_inputs(N, q)
        rng = np.random.default_rng(1)
        bits = pack_variants((rng.random((16384, N)) < rng.uniform(0.05, 0.95, 16384)[:, None]).astype(np.uint8))
        self.e = Engine(N); self.e.set_af_filter(0.01, 0.99); self.e.glm_setup(y, W, False, nl, nf)

        def run():
            while not self.stop:
                self.e.glm_batch(bits)
        self.th = threading.Thread(target=run); self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop = True; self.th.join(); self.e.close()


def _same(a, b):
    return all(np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(b[k]).view(np.uint8)) for k in a if isinstance(a[k], np.ndarray))


@pytest.mark.parametrize("N,q,V", [(4100, 1, 1024), (4099, 3, 777), (4100, 2, 1024)])
def test_forced_firth_calls_repeat_bit_for_bit(N, q, V, monkeypatch):
    from pyseer_amd.engine import Engine, pack_variants
    from pyseer_amd.model import fit_null
    rng = np.random.default_rng(97 + N + q)
    W = rng.standard_normal((N, q)); W /= np.abs(W).max(axis=0)
    eta = -0.3 + 1.2 * W[:, 0] - (0.7 * W[:, 1] if q > 1 else 0.0)
    y = (rng.random(N) < 1 / (1 + np.exp(-eta))).astype(float)
    af = np.concatenate([rng.uniform(0.02, 0.98, V - V // 4), rng.uniform(0.0101, 0.03, V // 8), rng.uniform(0.97, 0.9899, V // 4 - V // 8)])
    K = (rng.random((V, N)) < af[:, None]).astype(np.uint8)
    e0 = np.zeros((0, 0))
    nl = fit_null(y, W, e0, False).llf; nf = fit_null(y, W, e0, False, firth=True)
    bits = pack_variants(K)
    for route in (None, "firth_w=0", "firth_first32=0", "firth_first32=1"):
        if route is None:
            monkeypatch.delenv("SEERHIP_ROUTE", raising=False)
        else:
            monkeypatch.setenv("SEERHIP_ROUTE", route)
        e = Engine(N); e.set_af_filter(0.01, 0.99); e.glm_setup(y, W, False, nl, nf, force_firth=True)
        first = {k: np.array(v, copy=True) for k, v in e.glm_batch(bits).items() if isinstance(v, np.ndarray)}
        differ = 0
        with _Busy():
            for _ in range(16):
                differ += 0 if _same(first, e.glm_batch(bits)) else 1
        e.close()
        assert differ == 0, (route, differ)
    monkeypatch.delenv("SEERHIP_ROUTE", raising=False)


def test_lmm_and_logistic_calls_repeat_bit_for_bit():
    import torch
    import bench
    from pyseer_amd.engine import Engine, row_bytes_for
    dev = torch.device("cuda", 0)
    V = 1 << 15
    N = bench.N_SAMPLES
    U, S, h2, C, y, lin = bench.synth_lmm_inputs(N, 1003, dev)
    bits = bench.synth_bits(V, N, row_bytes_for(N), 77, dev)
    yy, W, nl, nf = bench.synth_glm_inputs(N, 10)
    for lmm in (True, False):
        e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
        if lmm:
            e.lmm_setup(U, S, y, C, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0)
        else:
            e.glm_setup(yy, W, False, nl, nf, 1.0, 1.0)
        nrow = 5 if lmm else 15
        run = e.lmm_batch_dev if lmm else e.glm_batch_dev
        out = torch.empty((nrow, V), dtype=torch.float64, device=dev); fl = torch.empty((V,), dtype=torch.int32, device=dev)
        run(bits, out, fl); torch.cuda.synchronize()
        first = (out.view(torch.int64).clone(), fl.clone())
        differ = 0
        with _Busy():
            for _ in range(10):
                run(bits, out, fl); torch.cuda.synchronize()
                differ += 0 if (torch.equal(first[0], out.view(torch.int64)) and torch.equal(first[1], fl)) else 1
        e.close()
        assert differ == 0, (lmm, differ)
