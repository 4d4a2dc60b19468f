"""Multi-GPU sharding of the k-mer stream: contiguous variant ranges per rank / per device, per-run constants
replicated, no collective on the data path.  Two forms: one PROCESS per GPU under torch.distributed (bench.py; the helpers below carry its
set-up broadcast and end-of-run reductions) and one process driving one context per GPU (the command line, pyseer_amd/__main__.py: one
pipelined stream per device over its own range of the packed cache; ShardedEngine: a batch split across contexts behind the Engine interface)."""
import numpy as np


def shard_bounds(n_items, rank, world):
    """Contiguous [lo, hi) of `n_items` for `rank`; sizes differ by at most one, order preserved across ranks."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# ---- set-up and end-of-run traffic of a one-process-per-GPU job (bench.py under torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" in the
# CPU tests).  `device`: where the collective's tensors live -- the rank's GPU for nccl, "cpu" for gloo.  None of this is on the data path.
def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def ranks_connected(device="cpu"):
    """How many ranks the collective library really connected: an all-reduce of ones (1 when torch.distributed is not initialised)."""
    import torch
    dist = _dist()
    if dist is None:
        return 1
    one = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(one)
    return int(one.item())


def broadcast_run_constants(arrays, src=0, device="cpu"):
    """The per-run constants of the job ({name: float64 ndarray}, e.g. U, S, y, h2 of the LMM: rank `src` decomposes the kinship once)
    -> the same dict on every rank.  Ranks other than `src` pass None.  Names and shapes travel first, then one broadcast per array."""
    import torch
    dist = _dist()
    if dist is None:
        return arrays
    me = dist.get_rank()
    meta = [[(k, tuple(np.asarray(v).shape)) for k, v in arrays.items()]] if me == src else [None]
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for name, shape in meta[0]:
        if me == src:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(arrays[name], dtype=np.float64).reshape(shape))).to(device)
        else:
            t = torch.empty(shape, dtype=torch.float64, device=device)
        dist.broadcast(t, src)
        out[name] = arrays[name] if me == src else t.cpu().numpy()
        del t
    return out


def max_over_ranks(x, device="cpu"):
    """MAX of a float over the ranks (the job's wall time is its slowest rank's)."""
    import torch
    dist = _dist()
    if dist is None:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(values, device="cpu"):
    """Every rank's small vector of floats, in rank order: list (one entry per rank) of lists."""
    import torch
    dist = _dist()
    if dist is None:
        return [[float(v) for v in values]]
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    g = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(g, mine)
    return [[float(v) for v in t.cpu()] for t in g]


class ShardedEngine(object):
    """One Engine per device, driven by one host thread each (the C ABI calls release the GIL); a batch of packed rows is
    split into contiguous shards (shard_bounds), results come back in input order.  The LMM's per-run constants are built once and copied
    device to device at set-up time (lmm_setup); the per-variant path has no device-to-device traffic at all."""

    def __init__(self, n_samples, devices):
        from .engine import Engine
        self.devices = list(devices)
        self.engines = [Engine(n_samples, device=d) for d in self.devices]

    def close(self):
        for e in self.engines:
            e.close()

    def _each(self, fn, retry=False):
        """fn(i, engine) on every context, one host thread each.  retry (the per-batch calls): a shard whose context failed is run once
        more on a context that succeeded -- shards are independent and every context holds the same per-run state, so the result is the
        same ("per-GPU shard failure -> re-run that shard"); the failed context is remembered in `failed_shards`.
        Set-up calls do not retry: their failure is the caller's to see."""
        import threading
        n = len(self.engines)
        out = [None] * n
        err = [None] * n

        def run(i, j):
            try:
                out[i] = fn(i, self.engines[j])
                err[i] = None
            except BaseException as ex:      # re-raised in the caller's thread
                err[i] = ex
        th = [threading.Thread(target=run, args=(i, i)) for i in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        bad = [i for i in range(n) if err[i] is not None]
        if bad and retry:
            good = [i for i in range(n) if err[i] is None]
            for i in list(bad):
                if not good:
                    break
                first = err[i]
                self.failed_shards = getattr(self, "failed_shards", []) + [(i, self.devices[i], repr(first))]
                run(i, good[len(self.failed_shards) % len(good)])
                if err[i] is None:
                    bad.remove(i)
        if bad:
            raise err[bad[0]]
        return out

    def set_af_filter(self, lo, hi):
        self._each(lambda i, e: e.set_af_filter(lo, hi))

    def lmm_setup(self, *a, **k):
        """The first context builds the per-run state (M = U~ diag(1/Sd) U~^T, limbs, tables); the others receive it device to device
        (sh_lmm_share: 94 MB at N = 5000) instead of each re-deriving it from the host copy of U."""
        self.engines[0].lmm_setup(*a, **k)
        for e in self.engines[1:]:
            e.lmm_share_from(self.engines[0])

    def glm_setup(self, *a, **k):
        self._each(lambda i, e: e.glm_setup(*a, **k))

    def _sharded(self, bits, method):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        world = len(self.engines)
        spans = [shard_bounds(bits.shape[0], r, world) for r in range(world)]
        parts = self._each(lambda i, e: getattr(e, method)(bits[spans[i][0]:spans[i][1]]), retry=True)
        return {k_: np.concatenate([p[k_] for p in parts], axis=0) for k_ in parts[0]}

    def lmm_batch(self, bits):
        return self._sharded(bits, "lmm_batch")

    def glm_batch(self, bits):
        return self._sharded(bits, "glm_batch")

    def set_dedup(self, on=True):
        """Per shard: a pattern repeated within one shard is tested once; results are identical either way."""
        self._each(lambda i, e: e.set_dedup(on))

    def lineage_setup(self, lin, cov=None):
        self._each(lambda i, e: e.lineage_setup(lin, cov))

    def lineage_batch(self, bits):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        world = len(self.engines)
        spans = [shard_bounds(bits.shape[0], r, world) for r in range(world)]
        return np.concatenate(self._each(lambda i, e: e.lineage_batch(bits[spans[i][0]:spans[i][1]]), retry=True), axis=0)
