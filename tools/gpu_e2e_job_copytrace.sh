#!/bin/bash
# usage (GPU box, after tools/gpu_e2e_job.py has written /tmp/e2e_job with E2E_MODEL=fixed): the fixed-effects job once more under
# rocprofv3 --kernel-trace --memory-copy-trace: how long the host-to-device copy of a block takes while the kernels of other blocks run
R=$GRAFT_REPO_ROOT; d=/tmp/e2e_job
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ct
env PYTHONPATH=$R ${ROUTE:+SEERHIP_ROUTE=$ROUTE} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ct -- python -m pyseer_amd --kmers $d/kmers.txt --uncompressed --phenotypes $d/pheno.tsv \
  --no-distances --covariates $d/cov.tsv --use-covariates 2q 3q 4q 5q 6q 7q 8q 9q 10q 11q --load-packed $d/kmers.seerpack --block_size 262144 --no-dedup --lrt-pvalue 1e-3 > /tmp/ct_out.tsv 2> /tmp/ct_err.txt
python - <<PY
import csv,glob
f=glob.glob('/tmp/ct/*/*memory_copy_trace.csv')
rows=list(csv.DictReader(open(f[0]))) if f else []
big=[r for r in rows if 'HOST_TO_DEVICE' in r['Direction'].upper()]
big=[r for r in big if int(r['End_Timestamp'])-int(r['Start_Timestamp']) > 1_000_000]
print('copies', len(rows), 'directions', sorted(set(r['Direction'] for r in rows)))
if big:
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in big]
    print('block copies (H2D, > 1 ms): %d, mean %.2f ms, min %.2f, max %.2f; at 165.7 MB per block: mean rate %.1f GB/s' % (len(d), sum(d)/len(d), min(d), max(d), 165.7/ (sum(d)/len(d))))
    t0=min(int(r['Start_Timestamp']) for r in big); t1=max(int(r['End_Timestamp']) for r in big)
    print('span of the block copies %.1f ms; busy %.1f ms' % ((t1-t0)/1e6, sum(d)))
k=glob.glob('/tmp/ct/*/*kernel_trace.csv')
kr=list(csv.DictReader(open(k[0])))
ours=[r for r in kr if 'k_glm' in r['Kernel_Name'] or 'k_firth' in r['Kernel_Name'] or 'k_job' in r['Kernel_Name'] or 'k_repack' in r['Kernel_Name'] or 'k_af' in r['Kernel_Name'] or 'k_bitdot' in r['Kernel_Name']]
t0=min(int(r['Start_Timestamp']) for r in ours); t1=max(int(r['End_Timestamp']) for r in ours)
# union of kernel busy intervals
iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in ours)
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print('kernels: span %.1f ms, device busy with them %.1f ms (%.0f %%)' % ((t1-t0)/1e6, busy/1e6, 100.0*busy/(t1-t0)))
PY
tail -3 /tmp/ct_err.txt | cut -c1-300
python - <<PY
# per batch (from its k_repack_bits to its k_job_scatter / last kernel on the same stream): stream, start, end relative to the first batch, ms
import csv,glob
k=glob.glob('/tmp/ct/*/*kernel_trace.csv')
kr=sorted(csv.DictReader(open(k[0])), key=lambda r:int(r['Start_Timestamp']))
bs=[r for r in kr if 'k_repack_bits' in r['Kernel_Name']]
t0=int(bs[0]['Start_Timestamp'])
by={}
for r in kr:
    by.setdefault(r['Stream_Id'],[]).append(r)
out=[]
for b in bs:
    sid=b['Stream_Id']; st=int(b['Start_Timestamp'])
    nxt=[int(x['Start_Timestamp']) for x in bs if x['Stream_Id']==sid and int(x['Start_Timestamp'])>st]
    lim=min(nxt) if nxt else 1<<62
    ks=[x for x in by[sid] if st<=int(x['Start_Timestamp'])<lim]
    en=max(int(x['End_Timestamp']) for x in ks)
    busy=sum(int(x['End_Timestamp'])-int(x['Start_Timestamp']) for x in ks)
    out.append((sid,(st-t0)/1e6,(en-t0)/1e6,busy/1e6,len(ks)))
for i,(sid,a,b,busy,n) in enumerate(out[8:24]):
    print('batch %2d stream %s  start %7.2f  end %7.2f  span %5.2f  kernel time %5.2f  kernels %d' % (i+8,sid,a,b,b-a,busy,n))
PY
python - <<PY
import csv,glob
k=glob.glob('/tmp/ct/*/*kernel_trace.csv')
kr=sorted(csv.DictReader(open(k[0])), key=lambda r:int(r['Start_Timestamp']))
bs=[r for r in kr if 'k_repack_bits' in r['Kernel_Name']]
t0=int(bs[0]['Start_Timestamp'])
b=bs[12]; sid=b['Stream_Id']; st=int(b['Start_Timestamp'])
nxt=[int(x['Start_Timestamp']) for x in bs if x['Stream_Id']==sid and int(x['Start_Timestamp'])>st]
lim=min(nxt) if nxt else 1<<62
prev=None
for x in kr:
    if x['Stream_Id']!=sid or not (st<=int(x['Start_Timestamp'])<lim): continue
    s_=int(x['Start_Timestamp']); e_=int(x['End_Timestamp'])
    gap=(s_-prev)/1e6 if prev else 0.0
    print('  %8.2f  +%6.2f gap  %6.3f ms  %s' % ((s_-t0)/1e6, gap, (e_-s_)/1e6, x['Kernel_Name'].replace('void ','').split('(')[0][:40]))
    prev=e_
mc=glob.glob('/tmp/ct/*/*memory_copy_trace.csv')
rows=sorted(csv.DictReader(open(mc[0])), key=lambda r:int(r['Start_Timestamp']))
print('H2D copies > 1 ms around it:')
for r in rows:
    s_=int(r['Start_Timestamp']); e_=int(r['End_Timestamp'])
    if e_-s_>1_000_000 and st-40_000_000 < s_ < lim: print('  copy %8.2f .. %8.2f' % ((s_-t0)/1e6,(e_-t0)/1e6))
PY
python - <<PY
# the device-level idle gaps (no kernel of ours running): the largest ones, with what ended before and what started after
import csv,glob,collections
k=glob.glob('/tmp/ct/*/*kernel_trace.csv')
kr=sorted(csv.DictReader(open(k[0])), key=lambda r:int(r['Start_Timestamp']))
ours=[r for r in kr if any(x in r['Kernel_Name'] for x in ('k_glm','k_firth','k_job','k_repack','k_af','k_bitdot','k_readback','rocclr'))]
t0=int(ours[0]['Start_Timestamp'])
gaps=[]; ce=int(ours[0]['End_Timestamp']); last=ours[0]
for r in ours[1:]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if s>ce: gaps.append((s-ce, ce, last, r))
    if e>ce: ce=e; last=r
tot=sum(g[0] for g in gaps)
print('device-level idle: %.1f ms in %d gaps; gaps > 0.2 ms: %.1f ms' % (tot/1e6, len(gaps), sum(g[0] for g in gaps if g[0]>200000)/1e6))
nm=lambda r: r['Kernel_Name'].replace('void ','').split('(')[0][:28]
byafter=collections.Counter(); bybefore=collections.Counter()
for g in gaps:
    byafter[nm(g[3])]+=g[0]; bybefore[nm(g[2])]+=g[0]
print('idle time by the kernel that STARTS after the gap:', [(k,round(v/1e6,1)) for k,v in byafter.most_common(6)])
print('idle time by the kernel that ENDED before the gap:', [(k,round(v/1e6,1)) for k,v in bybefore.most_common(6)])
for g in sorted(gaps, reverse=True)[:8]:
    print('  gap %.2f ms at %.1f: after %s (stream %s) -> %s (stream %s)' % (g[0]/1e6,(g[1]-t0)/1e6,nm(g[2]),g[2]['Stream_Id'],nm(g[3]),g[3]['Stream_Id']))
PY
