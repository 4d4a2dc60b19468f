"""Throughput of the native k-mer reader (csrc/reader.cpp) on a synthetic k-mer file, N samples x V k-mers, by container:
plain text, gzip through zlib's gzread (SEERHIP_ROUTE reader=zlib, the round-1 path), gzip through the in-tree inflate on one thread
(SEERHIP_ROUTE reader=serial) and on several (the default, csrc/inflate_par.h), BGZF (member-parallel).
No GPU involved; run it on the GPU host to see what feeds the engine there.  Prints one JSON line."""
import gzip, json, os, struct, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyseer_amd.input import NativeKmerReader

N = int(os.environ.get("N", 5000)); V = int(os.environ.get("V", 12000)); BS = int(os.environ.get("BLOCK", 4096))
d = os.environ.get("OUT", "/tmp/rb"); os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(0)
names = ["sample_%05d" % i for i in range(N)]
tok = np.array([n + ":1" for n in names], dtype=object)
t0 = time.time()
if os.environ.get("REUSE") and os.path.exists(d + "/k.bgzf.gz") and os.path.exists(d + "/meta.json") and json.load(open(d + "/meta.json")) == [N, V]:
    text = open(d + "/k.txt", "rb").read()                     # REUSE=1: the files of an earlier run with the same N, V
else:
    parts = []
    for v in range(V):
        af = rng.uniform(0.02, 0.98)
        idx = np.nonzero(rng.random(N) < af)[0]
        parts.append("".join(rng.choice(list("ACGT"), 31)) + " | " + " ".join(tok[idx]) + "\n")
    text = "".join(parts).encode(); del parts
    open(d + "/k.txt", "wb").write(text)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    with open(d + "/k.gz", "wb") as f:
        for i in range(0, len(text), 1 << 24):
            f.write(co.compress(text[i:i + (1 << 24)]))
        f.write(co.flush())
    with open(d + "/k.bgzf.gz", "wb") as f:
        for i in range(0, len(text), 65280):
            ch = text[i:i + 65280]
            c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(ch) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(comp) + 8 - 1))
            f.write(comp + struct.pack("<II", zlib.crc32(ch) & 0xFFFFFFFF, len(ch)))
    json.dump([N, V], open(d + "/meta.json", "w"))
gen = time.time() - t0
res = {"n_samples": N, "kmers": V, "text_MB": len(text) / 1e6, "gz_MB": os.path.getsize(d + "/k.gz") / 1e6, "cores": os.cpu_count(), "generate_s": gen}
want = None
TAGS = os.environ.get("TAGS", "").split(",") if os.environ.get("TAGS") else None
for tag, path, env in (("plain", "k.txt", None), ("gzip_zlib", "k.gz", "zlib"), ("gzip_fast", "k.gz", "serial"), ("gzip_par", "k.gz", None), ("bgzf", "k.bgzf.gz", None)):
    if TAGS and tag not in TAGS: continue
    base = ",".join(it for it in os.environ.get("SEERHIP_ROUTE", "").split(",") if it and not it.startswith("reader="))
    route = ",".join(x for x in (base, "reader=" + env if env else "") if x)
    if route: os.environ["SEERHIP_ROUTE"] = route
    else: os.environ.pop("SEERHIP_ROUTE", None)
    best = 0.0
    for rep in range(int(os.environ.get("REPS", 2))):
        t0 = time.time(); tot = 0; cs = 0
        for bits, counts, blob, off in NativeKmerReader(d + "/" + path, names, BS).raw_blocks():
            tot += counts.shape[0]; cs += int(counts.sum())
        dt = time.time() - t0
        best = max(best, tot / dt)
        assert tot == V
        if want is None: want = cs
        assert cs == want, (tag, cs, want)
    res[tag + "_kmers_per_s"] = best; res[tag + "_text_MBps"] = best * len(text) / V / 1e6
if TAGS:
    res["route"] = os.environ.get("SEERHIP_ROUTE")
    print(json.dumps(res)); sys.exit(0)
# the same gzip text cut into NF files at line boundaries and read as one stream (--kmers a.gz b.gz ...): one reader thread per file
import pandas as pd
from pyseer_amd.input import iter_packed_blocks_native_multi
NF = int(os.environ.get("FILES", 4))
lines = text.splitlines(True)
paths = []
for i in range(NF):
    path = d + "/part%d.gz" % i
    open(path, "wb").write(gzip.compress(b"".join(lines[i * len(lines) // NF:(i + 1) * len(lines) // NF]), 6))
    paths.append(path)
ph = pd.Series(np.zeros(N), index=names)
best = 0.0
for rep in range(2):
    t0 = time.time(); tot = 0
    for blk in iter_packed_blocks_native_multi(ph, paths, 0.0, 1.0, BS):
        tot += len(blk.names)
    best = max(best, tot / (time.time() - t0))
    assert tot == V
res["gzip_fast_%d_files_kmers_per_s" % NF] = best
from pyseer_amd.input import iter_packed_blocks_native
best = 0.0
for rep in range(2):                                        # the single file through the same block construction (AF filter, names), for comparison
    t0 = time.time(); tot = 0
    for blk in iter_packed_blocks_native(ph, d + "/k.gz", 0.0, 1.0, BS):
        tot += len(blk.names)
    best = max(best, tot / (time.time() - t0))
res["gzip_fast_1_file_blocks_kmers_per_s"] = best
print(json.dumps(res))
