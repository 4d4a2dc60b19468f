#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats, PMC passes.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/stats_bench.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --variants-per-step 262144 --no-cpu-baseline --no-extra > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --variants-per-step 262144 --no-cpu-baseline --no-extra > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --variants-per-step 262144 --no-cpu-baseline --no-extra > $O/pmc_sq.json 2> $O/pmc_sq.err
find $O -name "*.csv" | head -30
