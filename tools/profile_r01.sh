#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the same command,
# then the PMC passes (tools/profile_pmc.sh) and the summaries (tools/summarize_pmc.py) that are copied into profiles/r01/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > $O/stats_bench.json 2> $O/stats.err
bash $R/tools/profile_pmc.sh > $O/pmc.log 2>&1
python $R/tools/summarize_pmc.py $O $O/summary 262144
cat $O/bench_n1.json; cat $O/stats_bench.json
