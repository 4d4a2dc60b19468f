"""The same call again and again, with and without another stream keeping the device busy: LMM (C3's inputs), logistic (C2N5000's, C2's) and forced
Firth (C4's) on 2^16 rows of the bench's generator -- every output compared bit for bit with the first call's.  (The Firth shapes with few
covariates, where the race of profiles/r06/firth_determinism.txt showed, are in tools/gpu_firth_determinism.py.)"""
import json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for
REP = int(os.environ.get("REP", 25)); V = 1 << 16
dev = torch.device("cuda", 0)
stop = False


def busy():
    a = torch.randn(8192, 8192, device="cuda"); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(4):
                a = (a @ a).clamp(-1, 1)
            s.synchronize()


def check(name, run, nrow):
    global stop
    out = torch.empty((nrow, V), dtype=torch.float64, device=dev); fl = torch.empty((V,), dtype=torch.int32, device=dev)
    first = None; differ = 0
    for phase in (0, 1):
        if phase:
            stop = False; th = threading.Thread(target=busy); th.start()
        for r in range(REP):
            run(out, fl); torch.cuda.synchronize()
            cur = (out.view(torch.int64).clone(), fl.clone())
            if first is None:
                first = cur; continue
            if not (torch.equal(first[0], cur[0]) and torch.equal(first[1], cur[1])):
                differ += 1
        if phase:
            stop = True; th.join()
    r_ = {"what": name, "rows": V, "calls": 2 * REP, "calls_that_differ": differ}
    print(json.dumps(r_), flush=True)
    return differ


bad = 0
N = bench.N_SAMPLES
U, S, h2, C, y, lin = bench.synth_lmm_inputs(N, 1003, dev)
bits = bench.synth_bits(V, N, row_bytes_for(N), 77, dev)
e = Engine(N); e.use_torch_stream(); e.set_af_filter(0.01, 0.99); e.lmm_setup(U, S, y, C, h2, continuous=False, filter_pvalue=1.0, lrt_pvalue=1.0)
bad += check("LMM N=5000", lambda o, f: e.lmm_batch_dev(bits, o, f), 5); e.close()
for n_, firth in ((5000, False), (5000, True), (1000, False)):
    yy, W, nl, nf = bench.synth_glm_inputs(n_, 10)
    b = bench.synth_bits(V, n_, row_bytes_for(n_), 78, dev)
    e = Engine(n_); e.use_torch_stream(); e.set_af_filter(0.01, 0.99); e.glm_setup(yy, W, False, nl, nf, 1.0, 1.0, force_firth=firth)
    bad += check("%s N=%d q=10" % ("forced Firth" if firth else "logistic", n_), lambda o, f: e.glm_batch_dev(b, o, f), 15); e.close()
print("ANY_DIFFERENCE", bool(bad))
