"""bench.py's set-up broadcast and end-of-run reductions (pyseer_amd/parallel.py) over RCCL itself -- backend "nccl", tensors on the GPU -- with
the one rank a 1-GPU box has: the world_size-2 tests (tests/test_parallel_gloo.py, test_bench_ranks_on_one_gpu) run the same helpers over gloo
on CPU tensors, so without this the nccl branch (process group bound to a device, float64 / int32 collectives on device tensors, the object
broadcast of names and shapes) would first execute on the driver's multi-GPU run.  A subprocess: the process group is the process's.
The reference's counterpart is the pool of `--cpu N` workers (pyseer/__main__.py:541-568), which share nothing but their inputs."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from pyseer_amd import parallel as par
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert par.ranks_connected(dev) == 1 == dist.get_world_size()
rng = np.random.default_rng(5)
consts = {"U": rng.standard_normal((257, 129)), "S": rng.random(129), "h2": np.array([0.4737]), "empty": np.zeros((0, 3))}
got = par.broadcast_run_constants(consts, 0, dev)
assert list(got) == list(consts) and all(np.array_equal(got[k], consts[k]) for k in consts)
dist.barrier()
assert par.max_over_ranks(1.25, dev) == 1.25
assert par.gather_floats([3.5, -2.0], dev) == [[3.5, -2.0]]
dist.barrier(); dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
'''


def test_the_collective_helpers_over_rccl_with_one_rank():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
