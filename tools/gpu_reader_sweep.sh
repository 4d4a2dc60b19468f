#!/bin/bash
# Parallel gzip decoder: region size and translating threads (host only; run on the GPU box: 256 hardware threads under a quota of 16 CPUs).
mkdir -p gpurun_out/r04
export V=${V:-40000} OUT=/tmp/rb REPS=3
python tools/bench_reader.py > gpurun_out/r04/bench_reader.json 2> gpurun_out/r04/bench_reader.err
: > gpurun_out/r04/bench_reader_sweep.jsonl
for route in "reader_helpers=4" "reader_helpers=8" "reader_helpers=12" "reader_target=8e6,reader_chunk=4194304" "reader_target=8e6,reader_chunk=4194304,reader_helpers=8" "reader_target=16e6,reader_chunk=8388608,reader_helpers=8" "reader_helpers=8,reader_workers=8" "reader_helpers=6,reader_workers=6"; do
  REUSE=1 TAGS=gzip_par SEERHIP_ROUTE=$route python tools/bench_reader.py >> gpurun_out/r04/bench_reader_sweep.jsonl 2>> gpurun_out/r04/bench_reader.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/bench_reader_sweep.jsonl"):
    d = json.loads(l); print(d.get("route"), {k: round(v) for k, v in d.items() if k.endswith("kmers_per_s")})
PY
