"""Two engine contexts on ONE device, each fed its own resident batches from its own host thread (what `python -m pyseer_amd --gpus 0,0` does
for a job): does the second stream fill the gaps the per-variant kernels and the host round trips of one stream leave?  Prints one JSON line
per configuration: variants/s with one context and with two (and three)."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for

def run(cfg, nctx, steps=6, warmup=2):
    N = 1000 if cfg == "C2" else 5000
    q = 10
    Vs = (1 << 20) if cfg == "C2" else (1 << 18)
    rb = row_bytes_for(N)
    dev = torch.device("cuda:0")
    ctxs = []
    for c in range(nctx):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            if cfg == "C3":
                eng = Engine(N, device=0); eng.use_torch_stream()
                U, S, y, cov, h2 = bench.synth_lmm_inputs(N) if hasattr(bench, "synth_lmm_inputs") else (None,) * 5
            else:
                y, W, nl, nf = bench.synth_glm_inputs(N, q)
                eng = Engine(N, device=0); eng.use_torch_stream(); eng.set_af_filter(0.01, 0.99)
                eng.glm_setup(y, W, False, nl, nf, 1.0, 1.0, force_firth=(cfg == "C4"))
            bits = [bench.synth_bits(Vs, N, rb, 4242 + 10 * c + i, dev) for i in range(2)]
            out = torch.empty((5 + q, Vs), dtype=torch.float64, device=dev); fl = torch.empty((Vs,), dtype=torch.int32, device=dev)
        ctxs.append((st, eng, bits, out, fl))
    torch.cuda.synchronize()
    def work(c, n):
        st, eng, bits, out, fl = ctxs[c]
        with torch.cuda.stream(st):
            for i in range(n):
                eng.glm_batch_dev(bits[i % 2], out, fl)
            st.synchronize()
    def timed(n):
        th = [threading.Thread(target=work, args=(c, n)) for c in range(nctx)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    timed(warmup)
    dt = timed(steps)
    for c in ctxs: c[1].close()
    return Vs * steps * nctx / dt

res = {}
for cfg in (sys.argv[1:] or ["C2", "C2N5000", "C4"]):
    counts = [int(x) for x in os.environ.get("CONTEXTS", "1,2,3,1").split(",")]
    names = {1: "one", 2: "two", 3: "three", 4: "four", 5: "five", 6: "six", 8: "eight"}
    r = {}
    for k in counts:
        r[names.get(k, str(k)) + ("_again" if names.get(k, str(k)) in r else "")] = run(cfg, k, steps=6 if k < 4 else 4)
    res[cfg] = r
    print(json.dumps({cfg: {k: round(v / 1e6, 2) for k, v in r.items()}}), flush=True)
