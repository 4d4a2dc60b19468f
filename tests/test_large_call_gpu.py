"""Maximum sizes: ONE call of more than 2^22 rows (index arithmetic past 2^31 in the workspaces: rows x padded variants, the hand-over
workspace of the one-pass Firth kernel at 480 rows x 4.2 M fits, the LMM's limb partials) against the same rows handed over 2^18 at a time.
A variant's result does not depend on the batch it came in (the job stream and the lanes rest on that: tests/test_job_gpu.py), so the two must
agree BIT FOR BIT -- every output column and the flags.  LMM (C3's inputs), logistic and forced Firth (C2N5000 / C4's inputs) at N = 5000.
The reference has no counterpart: its loop takes a variant at a time (pyseer/__main__.py:541-593); `--block_size` only sets how many are read."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

V_BIG = (1 << 22) + 4099            # not a multiple of anything the kernels tile by
CHUNK = 1 << 18


def _bits(N, dev):
    import torch
    import bench
    from pyseer_amd.engine import row_bytes_for
    rb = row_bytes_for(N)
    parts = [bench.synth_bits(min(1 << 20, V_BIG - s), N, rb, 9100 + s, dev) for s in range(0, V_BIG, 1 << 20)]
    return torch.cat(parts)


def _compare(run, nrow, bits, dev):
    import torch
    out = torch.empty((nrow, V_BIG), dtype=torch.float64, device=dev); fl = torch.empty((V_BIG,), dtype=torch.int32, device=dev)
    run(bits, out, fl); torch.cuda.synchronize()
    bad = 0
    for s in range(0, V_BIG, CHUNK):
        n = min(CHUNK, V_BIG - s)
        o = torch.empty((nrow, n), dtype=torch.float64, device=dev); f = torch.empty((n,), dtype=torch.int32, device=dev)
        run(bits[s:s + n], o, f); torch.cuda.synchronize()
        bad += int((o.view(torch.int64) != out[:, s:s + n].view(torch.int64)).sum().item()) + int((f != fl[s:s + n]).sum().item())
    fitted = int(torch.isfinite(out[2]).sum().item())
    return bad, fitted


def test_one_lmm_call_of_four_million_rows_equals_its_chunks():
    import torch
    import bench
    from pyseer_amd.engine import Engine
    dev = torch.device("cuda", 0)
    N = bench.N_SAMPLES
    U, S, h2, C, y, lin = bench.synth_lmm_inputs(N, 1003, dev)
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.lmm_setup(U, S, y, C, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0)
    bits = _bits(N, dev)
    bad, fitted = _compare(lambda b, o, f: e.lmm_batch_dev(b, o, f), 5, bits, dev)
    e.close()
    assert bad == 0 and fitted > V_BIG // 2, (bad, fitted)


@pytest.mark.parametrize("firth", [False, True])
def test_one_fixed_effects_call_of_four_million_rows_equals_its_chunks(firth):
    import torch
    import bench
    from pyseer_amd.engine import Engine
    dev = torch.device("cuda", 0)
    N, q = bench.N_SAMPLES, 10
    y, W, nl, nf = bench.synth_glm_inputs(N, q)
    e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99)
    e.glm_setup(y, W, False, nl, nf, 1.0, 1.0, force_firth=firth)
    bits = _bits(N, dev)
    bad, fitted = _compare(lambda b, o, f: e.glm_batch_dev(b, o, f), 5 + q, bits, dev)
    e.close()
    assert bad == 0 and fitted > V_BIG // 2, (bad, fitted)
