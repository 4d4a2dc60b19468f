#!/bin/bash
# Reader throughput against the slab look-ahead (host only; run on the GPU box: 256 hardware threads under a quota of 16 CPUs).
mkdir -p gpurun_out/r04
export V=${V:-40000} OUT=/tmp/rb REPS=3
(lscpu | grep -E "Model name|Socket|NUMA|Thread|Core"; cat /sys/fs/cgroup/cpu.max; free -g | head -2) > gpurun_out/r04/host.txt 2>&1
python tools/bench_reader.py > gpurun_out/r04/bench_reader.json 2> gpurun_out/r04/bench_reader.err
: > gpurun_out/r04/bench_reader_sweep.jsonl
for d in 3 6 10 20; do
  REUSE=1 TAGS=gzip_par,bgzf SEERHIP_ROUTE=reader_depth=$d python tools/bench_reader.py >> gpurun_out/r04/bench_reader_sweep.jsonl 2>> gpurun_out/r04/bench_reader.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/bench_reader_sweep.jsonl"):
    d = json.loads(l); print(d.get("reader_threads"), d.get("route"), {k: round(v) for k, v in d.items() if k.endswith("kmers_per_s")})
PY
