#!/bin/bash
# usage (GPU box, after tools/gpu_e2e_job.py E2E_MODEL=fixed wrote /tmp/e2e_job): the fixed-effects job under rocprofv3 --hip-runtime-trace
# --kernel-trace: for the largest device-wide idle gaps, which HIP calls of which threads were in progress
R=$GRAFT_REPO_ROOT; d=/tmp/e2e_job
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/at
env PYTHONPATH=$R ${ROUTE:+SEERHIP_ROUTE=$ROUTE} rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/at -- python -m pyseer_amd --kmers $d/kmers.txt --uncompressed --phenotypes $d/pheno.tsv \
  --no-distances --covariates $d/cov.tsv --use-covariates 2q 3q 4q 5q 6q 7q 8q 9q 10q 11q --load-packed $d/kmers.seerpack --block_size 262144 --no-dedup --lrt-pvalue 1e-3 > /tmp/at_out.tsv 2> /tmp/at_err.txt
ls /tmp/at/*/ | head
python - <<PY
import csv,glob,collections
k=glob.glob('/tmp/at/*/*kernel_trace.csv')
kr=sorted(csv.DictReader(open(k[0])), key=lambda r:int(r['Start_Timestamp']))
ours=[r for r in kr if any(x in r['Kernel_Name'] for x in ('k_glm','k_firth','k_job','k_repack','k_af','k_bitdot','k_readback','rocclr'))]
gaps=[]; ce=int(ours[0]['End_Timestamp'])
for r in ours[1:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if s>ce: gaps.append((s-ce, ce, s))
    if e>ce: ce=e
t0=int(ours[0]['Start_Timestamp'])
steady=[g for g in gaps if (g[1]-t0)>150e6]
print('steady-state device idle: %.1f ms in %d gaps' % (sum(g[0] for g in steady)/1e6, len(steady)))
a=glob.glob('/tmp/at/*/*hip_api_trace.csv')
ar=list(csv.DictReader(open(a[0])))
print('api rows', len(ar), 'columns', list(ar[0].keys()))
# long API calls in steady state
calls=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']), r) for r in ar if int(r['Start_Timestamp'])-t0>150e6]
tot=collections.Counter(); cnt=collections.Counter(); mx=collections.Counter()
for dt,r in calls:
    tot[r['Function']]+=dt; cnt[r['Function']]+=1; mx[r['Function']]=max(mx[r['Function']],dt)
print('HIP calls in steady state by total time:')
for f,t in tot.most_common(14): print('  %-34s n %6d total %8.1f ms  mean %7.3f ms  max %7.2f ms' % (f,cnt[f],t/1e6,t/1e6/cnt[f],mx[f]/1e6))
# for the 6 largest steady gaps: calls overlapping the gap, by thread
for g in sorted(steady, reverse=True)[:6]:
    print('gap %.2f ms at %.1f ms:' % (g[0]/1e6,(g[1]-t0)/1e6))
    ov=[(r['Thread_Id'],r['Function'],(int(r['Start_Timestamp'])-g[1])/1e6,(int(r['End_Timestamp'])-g[1])/1e6) for dt,r in calls if int(r['Start_Timestamp'])<g[2] and int(r['End_Timestamp'])>g[1] and dt>100000]
    for o in ov[:10]: print('     thread %s %-28s from %+7.2f to %+7.2f ms (relative to the gap start)' % o)
PY
