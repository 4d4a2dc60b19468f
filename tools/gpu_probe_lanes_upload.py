"""Does a concurrent stream of host-to-device copies (what the job stream adds to the bench's resident rows) slow the fixed-effects batches?
One context, 3 lanes, C2N5000: variants/s with and without a side thread that uploads 166 MB pinned buffers back to back."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pyseer_amd.engine import Engine, row_bytes_for

N, q, Vs = 5000, 10, 1 << 18
rb = row_bytes_for(N); dev = torch.device("cuda:0")
y, W, nl, nf = bench.synth_glm_inputs(N, q)
eng = Engine(N, device=0); eng.use_torch_stream(); eng.set_af_filter(0.01, 0.99)
eng.glm_setup(y, W, False, nl, nf, 1.0, 1.0)
nb = 4
bits = [bench.synth_bits(Vs, N, rb, 4242 + i, dev) for i in range(nb)]
outs = [torch.empty((5 + q, Vs), dtype=torch.float64, device=dev) for _ in range(8)]
fls = [torch.empty((Vs,), dtype=torch.int32, device=dev) for _ in range(8)]
host = torch.empty((Vs, rb), dtype=torch.uint8).pin_memory()
dst = [torch.empty((Vs, rb), dtype=torch.uint8, device=dev) for _ in range(2)]
stop = threading.Event(); copied = [0]

def uploader(pace_s):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        i = 0
        while not stop.is_set():
            dst[i & 1].copy_(host, non_blocking=True); st.synchronize(); copied[0] += 1; i += 1
            if pace_s: time.sleep(pace_s)

def timed(steps):
    for i in range(6): eng.glm_batch_dev_async(bits[i % nb], outs[i % 8], fls[i % 8])
    eng.wait(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): eng.glm_batch_dev_async(bits[i % nb], outs[i % 8], fls[i % 8])
    eng.wait(); torch.cuda.synchronize()
    return Vs * steps / (time.perf_counter() - t0) / 1e6

res = {"no_uploads": timed(24)}
for name, pace in (("uploads_back_to_back", 0.0), ("uploads_one_per_7ms", 0.004)):
    stop.clear(); copied[0] = 0
    th = threading.Thread(target=uploader, args=(pace,)); th.start()
    time.sleep(0.05); c0 = copied[0]; t0 = time.perf_counter()
    res[name] = timed(24)
    dt = time.perf_counter() - t0; res[name + "_GBps"] = (copied[0] - c0) * Vs * rb / dt / 1e9
    stop.set(); th.join()
print(json.dumps({k: round(v, 2) for k, v in res.items()}))
