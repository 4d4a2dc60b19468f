#!/bin/bash
# after tools/gpu_e2e_c3.py (same gpurun call: its inputs stay in /tmp/e2e_c3): one overlapped CLI run with the engine's per-chunk host timing
d=/tmp/e2e_c3
SEERHIP_DEBUG=host,cli PYTHONPATH=. python -m pyseer_amd --kmers $d/kmers.txt --uncompressed --phenotypes $d/pheno.tsv --lmm \
  --load-lmm $d/lmm.npz --load-packed $d/kmers.seerpack --block_size ${BLOCK:-262144} --no-dedup > $d/out_dbg.tsv 2> $d/dbg.err
grep "host_batch" $d/dbg.err | sed -n '20,32p'
grep "cli timing" $d/dbg.err
