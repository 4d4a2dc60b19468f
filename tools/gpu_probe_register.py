"""Where does the CLI's registration time go?  Python mmap of a file in the page cache, windows as pyseer_amd/input.py cuts them, each
registered (sh_host_register), read by a Job (LMM at N = 5000 would need a set-up: a plain H2D through torch is enough here) and unregistered;
wall and thread-CPU time of each call."""
import mmap, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pyseer_amd import _abi
lib = _abi.load()
path = "/tmp/probe_register.bin"
S = 2 << 30; W = 166 << 20
with open(path, "wb") as f:
    blk = np.random.default_rng(0).integers(0, 255, 1 << 24, dtype=np.uint8).tobytes()
    for _ in range(S >> 24):
        f.write(blk)
    f.flush(); os.fsync(f.fileno())
torch.zeros(1, device="cuda")
dst = torch.empty(W, dtype=torch.uint8, device="cuda")
A = torch.randn(8192, 8192, device="cuda"); side = torch.cuda.Stream()
def load(n=12):
    with torch.cuda.stream(side):
        for _ in range(n):
            torch.mm(A, A)
for variant in ("plain", "under_load", "under_load_from_thread", "populate"):
    with open(path, "rb") as f:
        flags = mmap.MAP_SHARED | (mmap.MAP_POPULATE if variant == "populate" else 0)
        mm = mmap.mmap(f.fileno(), 0, flags=flags, prot=mmap.PROT_READ)
    if variant == "madvise_sequential":
        mm.madvise(mmap.MADV_SEQUENTIAL)
    treg = tun = tcopy = creg = cun = ttouch = 0.0
    n = 0
    for o in range(0, S - W, W):
        a = np.frombuffer(mm, dtype=np.uint8, count=W, offset=o + 12345)
        if variant == "touch_first":
            t0 = time.perf_counter(); int(a[::4096].sum()); ttouch += time.perf_counter() - t0
        if variant.startswith("under_load"):
            load()
        if variant == "under_load_from_thread":
            import threading
            box = {}
            def reg():
                t0 = time.perf_counter(); c0 = time.thread_time()
                box["rc"] = lib.sh_host_register(a.ctypes.data, a.nbytes, 0)
                box["t"] = time.perf_counter() - t0; box["c"] = time.thread_time() - c0
            th = threading.Thread(target=reg); th.start(); th.join()
            rc = box["rc"]; treg += box["t"]; creg += box["c"]
        else:
            t0 = time.perf_counter(); c0 = time.thread_time()
            rc = lib.sh_host_register(a.ctypes.data, a.nbytes, 0)
            treg += time.perf_counter() - t0; creg += time.thread_time() - c0
        assert rc == 0, lib.sh_last_error()
        t0 = time.perf_counter()
        src = torch.from_numpy(a)
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        tcopy += time.perf_counter() - t0
        if variant.startswith("under_load"):
            load(4)
        t0 = time.perf_counter(); c0 = time.thread_time()
        rc = lib.sh_host_unregister(a.ctypes.data)
        tun += time.perf_counter() - t0; cun += time.thread_time() - c0
        assert rc == 0, lib.sh_last_error()
        n += 1
    rows = n * W / 632.0
    print("%-20s %d windows: register %.1f ms each (%.4f CPU-s per M rows), unregister %.1f ms each (%.4f), copy %.1f ms each (%.1f GB/s), touch %.1f ms each"
          % (variant, n, treg / n * 1e3, creg / rows * 1e6, tun / n * 1e3, cun / rows * 1e6, tcopy / n * 1e3, W / (tcopy / n) / 1e9, ttouch / n * 1e3), flush=True)
    torch.cuda.synchronize()
    del a, src
    mm.close()
os.remove(path)
