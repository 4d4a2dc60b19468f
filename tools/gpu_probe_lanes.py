"""The lanes of ONE engine context (sh_glm_batch_dev_async, csrc/lanes_api.inc): variants/s of the fixed-effects bench configurations with
1, 2, 3, 4 lanes against the synchronous call, and the check that a lane's rows are the synchronous call's bit for bit.
usage (GPU box): python tools/gpu_probe_lanes.py [C2N5000 C4 C2]   (LANES=1,2,3,4  STEPS=12)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for


def run(cfg, lanes_list, steps):
    N = 1000 if cfg == "C2" else 5000
    q = 10
    Vs = (1 << 20) if cfg == "C2" else (1 << 18)
    rb = row_bytes_for(N)
    dev = torch.device("cuda:0")
    y, W, nl, nf = bench.synth_glm_inputs(N, q)
    eng = Engine(N, device=0); eng.use_torch_stream(); eng.set_af_filter(0.01, 0.99)
    eng.glm_setup(y, W, False, nl, nf, 1.0, 1.0, force_firth=(cfg == "C4"))
    nb = 4
    bits = [bench.synth_bits(Vs, N, rb, 4242 + i, dev) for i in range(nb)]
    outs = [torch.empty((5 + q, Vs), dtype=torch.float64, device=dev) for _ in range(nb)]
    fls = [torch.empty((Vs,), dtype=torch.int32, device=dev) for _ in range(nb)]
    ref_out = [torch.empty_like(outs[0]) for _ in range(nb)]; ref_fl = [torch.empty_like(fls[0]) for _ in range(nb)]
    for i in range(nb):
        eng.glm_batch_dev(bits[i], ref_out[i], ref_fl[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.glm_batch_dev(bits[i % nb], outs[i % nb], fls[i % nb])
    torch.cuda.synchronize()
    res = {"sync": Vs * steps / (time.perf_counter() - t0) / 1e6}
    for L in lanes_list:
        eng.set_lanes(L)
        for i in range(max(L, 2)):
            eng.glm_batch_dev_async(bits[i % nb], outs[i % nb], fls[i % nb])
        eng.wait(); torch.cuda.synchronize()
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.glm_batch_dev_async(bits[i % nb], outs[i % nb], fls[i % nb])
        eng.wait(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(bool(torch.equal(outs[i].view(torch.int64), ref_out[i].view(torch.int64))) and bool(torch.equal(fls[i], ref_fl[i])) for i in range(nb))
        res["lanes_%d" % L] = Vs * steps / dt / 1e6
        res["lanes_%d_identical" % L] = same
    eng.close()
    return res


for cfg in (sys.argv[1:] or ["C2N5000", "C4", "C2"]):
    lanes = [int(x) for x in os.environ.get("LANES", "1,2,3,4").split(",")]
    r = run(cfg, lanes, int(os.environ.get("STEPS", "12")))
    print(json.dumps({cfg: {k: (round(v, 2) if not isinstance(v, bool) else v) for k, v in r.items()}}), flush=True)
