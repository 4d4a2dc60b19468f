#!/bin/bash
# Reader throughput against the worker counts (host only; run on the GPU box for its 256 threads).
mkdir -p gpurun_out/r04
export V=${V:-40000} OUT=/tmp/rb REPS=3
(lscpu | grep -E "Model name|Socket|NUMA|Thread|Core"; cat /sys/fs/cgroup/cpu.max; free -g | head -2) > gpurun_out/r04/host.txt 2>&1
python tools/bench_reader.py > gpurun_out/r04/bench_reader.json 2> gpurun_out/r04/bench_reader.err
: > gpurun_out/r04/bench_reader_sweep.jsonl
for t in 16 24 32 48 64; do
  REUSE=1 TAGS=plain,gzip_par,bgzf SEERHIP_READER_THREADS=$t python tools/bench_reader.py >> gpurun_out/r04/bench_reader_sweep.jsonl 2>> gpurun_out/r04/bench_reader.err
done
for w in 12 24 48; do
  REUSE=1 TAGS=gzip_par SEERHIP_ROUTE=reader_workers=$w python tools/bench_reader.py >> gpurun_out/r04/bench_reader_sweep.jsonl 2>> gpurun_out/r04/bench_reader.err
  REUSE=1 TAGS=gzip_par SEERHIP_READER_THREADS=24 SEERHIP_ROUTE=reader_workers=$w python tools/bench_reader.py >> gpurun_out/r04/bench_reader_sweep.jsonl 2>> gpurun_out/r04/bench_reader.err
done
cat gpurun_out/r04/host.txt
python - <<'PY'
import json
for l in open("gpurun_out/r04/bench_reader_sweep.jsonl"):
    d = json.loads(l); print(d.get("reader_threads"), d.get("route"), {k: round(v) for k, v in d.items() if k.endswith("kmers_per_s")})
PY
