"""CPU baseline of the LMM block test, run the way the reference runs it: `--cpu P` worker processes (multiprocessing Pool,
pyseer/__main__.py:541-568, 777-780), each fitting whole blocks of --block_size variants with single-threaded BLAS
(docs/multiprocessing.rst: one BLAS thread per worker).  The arithmetic is oracle/lmm_blas.py (fit_lmm_block restated with numpy).

TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only).  Runs as its own process so that the workers are forked from an interpreter
that never touched the GPU runtime:

    python -m oracle.cpu_baseline_lmm inputs.npz [procs] [block] [blocks_per_proc]

inputs.npz holds U, S, y, C, h2.  Prints one JSON object: {"variants": n, "seconds": t, "procs": P, "block": b}.
"""
import json
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"                       # before numpy is imported: one BLAS thread per worker

import numpy as np  # noqa: E402

_L = None
_H2 = None
_BAR = None


def _warm(args):
    seed, block, n = args
    rng = np.random.default_rng(seed)
    Kv = (rng.random((block, n)) < rng.uniform(0.02, 0.98, block)[:, None]).astype(np.float64)
    _L.block(_H2, Kv)
    return 0.0


def _work(args):
    """One task per worker: draw all of this worker's blocks FIRST, meet the other workers at a barrier, then fit them; returns the time
    of the fits alone (round 2 drew the rows inside the timed region: ~10 % in the CPU's disfavour)."""
    seed, block, n, per = args
    rng = np.random.default_rng(seed)
    blocks = []
    for _ in range(per):
        af = rng.uniform(0.02, 0.98, block)
        blocks.append((rng.random((block, n)) < af[:, None]).astype(np.float64))
    _BAR.wait()
    t0 = time.time()
    for Kv in blocks:
        _L.block(_H2, Kv)
    return time.time() - t0


def main():
    global _L, _H2, _BAR
    import multiprocessing as mp
    from oracle.lmm_blas import LmmBlas
    d = np.load(sys.argv[1])
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, (os.cpu_count() or 2) // 2)
    block = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    per = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    _L = LmmBlas(d["U"], d["S"], d["y"], d["C"]); _H2 = float(d["h2"])
    n = _L.n
    ctx = mp.get_context("fork")               # the workers share U (copy-on-write), as the reference's Pool shares its LMM object
    _BAR = ctx.Barrier(procs)
    with ctx.Pool(procs) as pool:
        pool.map(_warm, [(1000 + i, 64, n) for i in range(procs)])           # start the workers, touch the pages
        # all workers start their fits together (the barrier), the job is as long as its slowest worker
        dts = pool.map(_work, [(i, block, n, per) for i in range(procs)], chunksize=1)
        dt = max(dts)
    print(json.dumps({"variants": procs * per * block, "seconds": dt, "procs": procs, "block": block}))


if __name__ == "__main__":
    main()
