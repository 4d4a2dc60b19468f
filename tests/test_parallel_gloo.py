"""N>1 path on CPU: world_size-2 gloo processes shard a variant range, 'test' their shard with a stand-in statistic,
and the gathered result / counters equal the single-process answer."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from pyseer_amd.parallel import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 1000, 12345):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    """What bench.py does around its timed region under torch.distributed, on CPU tensors over gloo: the rank count check, the set-up broadcast
    of the per-run constants from rank 0, every rank testing its own contiguous shard, MAX of the wall time, per-rank values in rank order."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pyseer_amd import parallel as par
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = par.ranks_connected("cpu")
    consts = None
    if rank == 0:
        rng = np.random.default_rng(3)
        consts = {"K": (rng.random((101, 40)) < 0.3).astype(np.float64), "h2": np.array([0.25]), "empty": np.zeros((0, 3))}
    consts = par.broadcast_run_constants(consts, 0, "cpu")
    K = consts["K"]
    lo, hi = par.shard_bounds(K.shape[0], rank, world)
    local = K[lo:hi].sum(axis=1)                              # stand-in for a per-variant statistic
    slowest = par.max_over_ranks(1.0 + rank, "cpu")
    g = par.gather_floats([float(hi - lo), float(local.sum()), float(consts["h2"][0])], "cpu")
    if rank == 0:
        q.put((seen, slowest, g, K.sum(), tuple(consts["empty"].shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    seen, slowest, g, total, eshape = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert seen == 2 and slowest == 2.0 and eshape == (0, 3)
    assert [x[0] for x in g] == [51.0, 50.0] and sum(x[1] for x in g) == total and [x[2] for x in g] == [0.25, 0.25]


def test_helpers_without_a_process_group_are_the_identity():
    from pyseer_amd import parallel as par
    assert par.ranks_connected() == 1 and par.max_over_ranks(3.5) == 3.5 and par.gather_floats([1, 2]) == [[1.0, 2.0]]
    d = {"a": np.arange(3.0)}
    assert par.broadcast_run_constants(d) is d


def test_a_failed_shard_is_rerun_on_another_context():
    """shards are independent, every context holds the same per-run state -> a shard whose context fails is run once more
    on a healthy one and the batch comes back complete and in order; a shard that fails twice raises.  (Stub engines: no GPU here.)"""
    import numpy as np
    import pytest
    from pyseer_amd.parallel import ShardedEngine

    class Stub(object):
        def __init__(self, fail_times):
            self.fail_times, self.calls = fail_times, 0

        def lmm_batch(self, bits):
            self.calls += 1
            if self.fail_times > 0:
                self.fail_times -= 1
                raise RuntimeError("device lost")
            return {"beta": bits[:, 0].astype(float) * 2.0, "flags": np.zeros(bits.shape[0], dtype=np.uint32)}

    se = object.__new__(ShardedEngine)
    se.devices = [0, 1, 2]
    se.engines = [Stub(0), Stub(1), Stub(0)]
    bits = np.arange(10, dtype=np.uint8).reshape(10, 1).repeat(8, axis=1)
    r = se.lmm_batch(bits)
    assert np.array_equal(r["beta"], np.arange(10) * 2.0)                   # complete, in input order
    assert len(se.failed_shards) == 1 and se.failed_shards[0][0] == 1
    assert se.engines[1].calls == 1 and se.engines[0].calls + se.engines[2].calls == 3
    r = se.lmm_batch(bits)                                                  # the context recovered: nothing to retry
    assert np.array_equal(r["beta"], np.arange(10) * 2.0) and len(se.failed_shards) == 1
    se.engines = [Stub(5), Stub(5)]; se.devices = [0, 1]
    with pytest.raises(RuntimeError):
        se.lmm_batch(bits)
