#!/bin/bash
# after tools/gpu_e2e_c3.py and tools/gpu_e2e_c2n5000.py (same gpurun call): one fixed-effects CLI run with the engine's per-chunk host timing
s=/tmp/e2e_c3; d=/tmp/e2e_c2
SEERHIP_DEBUG=host,cli PYTHONPATH=. python -m pyseer_amd --kmers $s/kmers.txt --uncompressed --phenotypes $d/pheno.tsv --no-distances \
  --covariates $d/cov.tsv --use-covariates 2q 3q 4q 5q 6q 7q 8q 9q 10q 11q --load-packed $s/kmers.seerpack --block_size ${BLOCK:-262144} --no-dedup > $d/out_dbg.tsv 2> $d/dbg.err
grep "host_batch" $d/dbg.err | sed -n "30,42p"
grep "cli timing" $d/dbg.err
