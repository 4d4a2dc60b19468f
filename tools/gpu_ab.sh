#!/bin/bash
# usage (on the GPU box): tools/gpu_ab.sh <config> <name> [<name> ...]   -- per-kernel averages of bench.py --config <config> for A/B builds in pyseer_amd/ab/
# ("cur" = the in-tree library).  Extra environment (e.g. SEERHIP_ROUTE=firth_literal=1) is passed through.
R=$GRAFT_REPO_ROOT; cfg=$1; shift
cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  lib=$R/pyseer_amd/ab/libseerhip_$n.so; [ "$n" = "cur" ] && lib=$R/pyseer_amd/libseerhip.so
  rm -rf /tmp/ab_$n
  SEERHIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$n -- python $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-parity > /tmp/ab_$n.json 2>/tmp/ab_$n.err
  echo "== $n: $(python -c "import json;d=json.loads(open('/tmp/ab_$n.json').readlines()[-1]);print('%.2f ms/step, %.3g variants/s' % (d['ms_per_step'], d['value']))" 2>/dev/null || tail -2 /tmp/ab_$n.err)"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/ab_$n/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:40]:
    nm=r['Name']
    if nm.startswith('void '): nm=nm[5:]
    if not (nm.startswith('k_') ): continue
    if float(r['TotalDurationNs'])/4e6 < 0.05: continue
    print('   %-34s calls %4s  %8.3f ms/step  avg %7.3f  max %7.3f' % (nm.split('(')[0][:34], r['Calls'], float(r['TotalDurationNs'])/4e6, float(r['AverageNs'])/1e6, float(r['MaxNs'])/1e6))
PY
done
