#!/bin/bash
# usage (on the GPU box): tools/kstats.sh <config> [route]  -- rocprofv3 kernel stats of bench.py --config <config> on ONE lane: name, calls, average and maximum duration (ms)
R=$GRAFT_REPO_ROOT; cfg=${1:-C2N5000}; route=${2:-lanes=1}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks
SEERHIP_ROUTE=$route rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-parity > /dev/null 2>&1
python - <<'PY'
import csv
for r in list(csv.DictReader(open('/tmp/ks/ks_kernel_stats.csv')))[:14]:
    n = r['Name'].replace('void ', '').split('(')[0]
    if n.startswith('at::'): continue
    print('%-34s calls %4s  avg %8.3f ms  max %8.3f ms  total %8.3f ms' % (n[:34], r['Calls'], float(r['AverageNs']) / 1e6, float(r['MaxNs']) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY
