"""The job stream driven directly (no command line, no reader): C2N5000's workload, blocks of 2^18 rows in pinned host memory, submit / collect
as pyseer_amd/__main__.py run_stream_job does it.  Rows/s with 1 and 3 lanes, against the engine's own rate on resident rows."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pyseer_amd.engine import Engine, Job, row_bytes_for
from pyseer_amd.sink import names_blob

N, q, Vs = 5000, 10, 1 << 18
rb = row_bytes_for(N); dev = torch.device("cuda:0")
y, W, nl, nf = bench.synth_glm_inputs(N, q)
NB = int(os.environ.get("BLOCKS", 40))
host = []
for i in range(4):
    t = bench.synth_bits(Vs, N, rb, 4242 + i, dev).cpu().pin_memory()
    host.append(t)
counts = [np.unpackbits(h.numpy(), axis=1).sum(axis=1).astype(np.int32) for h in host]
blob, off = names_blob(["K%07d" % v for v in range(Vs)])
res = {}
for lanes in (1, 3):
    eng = Engine(N, device=0); eng.set_af_filter(0.01, 0.99); eng.set_lanes(lanes)
    eng.glm_setup(y, W, False, nl, nf, 1.0, 1e-3)
    job = Job(eng, False)
    def run(nblocks):
        printed = 0; inflight = 0
        t0 = time.perf_counter()
        for k in range(nblocks):
            while job.pending() >= job.depth:
                _, c, _ = job.collect(); printed += c[2]
            job.submit(host[k % 4].numpy(), counts[k % 4], blob, off, rows_are_dma=True)
            while job.pending() > job.depth - 1:
                _, c, _ = job.collect(); printed += c[2]
        while job.pending():
            _, c, _ = job.collect(); printed += c[2]
        return Vs * nblocks / (time.perf_counter() - t0) / 1e6, printed
    run(8)
    r, p = run(NB)
    res["job_lanes_%d" % lanes] = round(r, 2); res["printed_%d" % lanes] = p
    job.close(); eng.close()
print(json.dumps(res))
