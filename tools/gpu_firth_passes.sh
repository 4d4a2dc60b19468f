#!/bin/bash
# usage (on the GPU box): tools/gpu_firth_passes.sh [<ab name>]  -- every Firth kernel dispatch of one bench step of C4, in order, with its duration
R=$GRAFT_REPO_ROOT; n=${1:-cur}
lib=$R/pyseer_amd/ab/libseerhip_$n.so; [ "$n" = "cur" ] && lib=$R/pyseer_amd/libseerhip.so
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/fp_$n
SEERHIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d /tmp/fp_$n -- python $R/bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /tmp/fp_$n.json 2>/tmp/fp_$n.err
echo "== $n: $(python -c "import json;d=json.loads(open('/tmp/fp_$n.json').readlines()[-1]);print('%.2f ms/step, %.3g variants/s' % (d['ms_per_step'], d['value']))" 2>/dev/null || tail -2 /tmp/fp_$n.err)"
python - <<PY
import csv,glob
f=glob.glob('/tmp/fp_$n/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ks=[(r['Kernel_Name'].replace('void ','').split('(')[0], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6) for r in rows]
# the last step: from the last k_glm_fast / first firth kernel group
idx=[i for i,(k,_) in enumerate(ks) if k.startswith('k_firth_fast')]
# split into steps by gaps of non-firth kernels: take the last 1/3
last=[]
i=len(ks)-1
while i>=0 and not ks[i][0].startswith('k_firth'): i-=1
while i>=0 and (ks[i][0].startswith('k_firth') or ks[i][1]<0.05): last.append(ks[i]); i-=1
for k,t in reversed(last):
    if t>=0.02: print('   %-28s %8.3f ms' % (k[:28], t))
PY
