#!/bin/bash
# usage (GPU box, after `E2E_MODEL=fixed python tools/gpu_e2e_job.py` has written /tmp/e2e_job): the fixed-effects job (logistic, 10 covariates)
# once more under rocprofv3 --kernel-trace --memory-copy-trace: the device's time between the first and the last sample pass of the block
# loop (union of busy intervals, per kernel totals, the copies) -- what separates the job's rows/s from the engine's rate on resident rows.
# BLOCK=<rows per block> ROUTE=<SEERHIP_ROUTE>
R=$GRAFT_REPO_ROOT; d=/tmp/e2e_job; B=${BLOCK:-262144}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ct
env PYTHONPATH=$R SEERHIP_DEBUG=cli ${ROUTE:+SEERHIP_ROUTE=$ROUTE} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ct -- python -m pyseer_amd --kmers $d/kmers.txt --uncompressed --phenotypes $d/pheno.tsv \
  --no-distances --covariates $d/cov.tsv --use-covariates 2q 3q 4q 5q 6q 7q 8q 9q 10q 11q --load-packed $d/kmers.seerpack --block_size $B --no-dedup --lrt-pvalue 1e-3 > /tmp/ct_out.tsv 2> /tmp/ct_err.txt
echo "== block $B ${ROUTE:+route $ROUTE}"; grep -h "cli budget\|rows/s" /tmp/ct_err.txt | tail -3 | cut -c1-400
python - <<PY
import csv,glob,collections
k=glob.glob('/tmp/ct/*/*kernel_trace.csv')
kr=list(csv.DictReader(open(k[0])))
ours=[r for r in kr if r['Kernel_Name'].startswith(('k_','void k_'))]
blk=[r for r in ours if 'k_glm_ll' in r['Kernel_Name']]
t0=min(int(r['Start_Timestamp']) for r in blk); t1=max(int(r['End_Timestamp']) for r in blk)
loop=[r for r in ours if t0<=int(r['Start_Timestamp'])<=t1]
iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in loop)
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print('block loop on the device: %.1f ms from the first to the last likelihood pass, %d of them, device busy %.1f ms (%.1f %%)' % ((t1-t0)/1e6, len(blk), busy/1e6, 100*busy/(t1-t0)))
tot=collections.Counter(); n=collections.Counter()
for r in loop:
    nm=r['Kernel_Name'].replace('void ','').split('(')[0]; tot[nm]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); n[nm]+=1
print('   (kernels of different lanes overlap: the sum of their durations exceeds the span)')
for nm,t in tot.most_common(16): print('   %-44s %5d launches %9.2f ms  %5.1f %% of the span' % (nm[:44], n[nm], t/1e6, 100*t/(t1-t0)))
f=glob.glob('/tmp/ct/*/*memory_copy_trace.csv')
rows=list(csv.DictReader(open(f[0]))) if f else []
big=[r for r in rows if int(r['End_Timestamp'])-int(r['Start_Timestamp']) > 1_000_000]
if big:
    dd=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in big]
    print('block copies (> 1 ms): %d, mean %.2f ms, max %.2f' % (len(dd), sum(dd)/len(dd), max(dd)))
PY
