"""Multi-GPU sharding of the k-mer stream: contiguous variant ranges per rank, per-run constants
replicated, no collective on the data path.  The only cross-rank traffic is the end-of-run sum of the driver's four
counters (loaded / pre-filtered / tested / printed, pyseer/__main__.py:831-834)."""
import numpy as np


def shard_bounds(n_items, rank, world):
    """Contiguous [lo, hi) of `n_items` for `rank`; sizes differ by at most one, order preserved across ranks."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sum_counters(counters, group=None):
    """All-reduce (SUM) of small integer counters across ranks; identity when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.asarray(counters, dtype=np.int64))
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def gather_in_order(local_rows, group=None):
    """Concatenate per-rank result arrays in rank order (= input order, which the reference guarantees even with
    --cpu N since starmap preserves order, pyseer/__main__.py:541-568)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(local_rows)
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, np.asarray(local_rows), group=group)
    return np.concatenate(parts, axis=0)
