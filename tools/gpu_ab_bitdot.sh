#!/bin/bash
# A/B of the carrier sums: int8 matrix cores (default) against round 3's nibble-table kernel (SEERHIP_ROUTE=bitdot=2), same box.
mkdir -p gpurun_out/r04
: > gpurun_out/r04/ab_bitdot.jsonl
for cfg in C2N5000 C2 C4; do
  for route in "" "bitdot=2" "" "bitdot=2"; do
    SEERHIP_ROUTE=$route python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'config':'$cfg','route':'$route','value':d['value'],'ms_per_step':d['ms_per_step']}))" >> gpurun_out/r04/ab_bitdot.jsonl
  done
done
cat gpurun_out/r04/ab_bitdot.jsonl
