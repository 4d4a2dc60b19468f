#!/bin/bash
# usage (GPU box, after tools/gpu_e2e_job.py has written /tmp/e2e_job; MODEL=fixed|lmm): the job once more under rocprofv3 --hip-trace --kernel-trace:
# which runtime calls fill the time between the start of the block loop and its steady state (allocations, registrations, first launches).
R=$GRAFT_REPO_ROOT; d=/tmp/e2e_job
if [ "${MODEL:-fixed}" = fixed ]; then M="--no-distances --covariates $d/cov.tsv --use-covariates 2q 3q 4q 5q 6q 7q 8q 9q 10q 11q"; else M="--lmm --load-lmm $d/lmm.npz"; fi
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/rt
env PYTHONPATH=$R SEERHIP_DEBUG=cli rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/rt -- python -m pyseer_amd --kmers $d/kmers.txt --uncompressed --phenotypes $d/pheno.tsv \
  $M --load-packed $d/kmers.seerpack --no-dedup --lrt-pvalue 1e-3 > /tmp/rt_out.tsv 2> /tmp/rt_err.txt
grep "job_run" /tmp/rt_err.txt | head -6; grep "cli timing" /tmp/rt_err.txt | cut -c1-200
python - <<PY
import csv,glob,collections
a=glob.glob('/tmp/rt/*/*hip_api_trace.csv'); k=glob.glob('/tmp/rt/*/*kernel_trace.csv')
api=list(csv.DictReader(open(a[0]))); kr=list(csv.DictReader(open(k[0])))
name='k_glm_ll' if '${MODEL:-fixed}'=='fixed' else 'quadform'
blk=sorted(int(r['Start_Timestamp']) for r in kr if name in r['Kernel_Name'])
# the loop starts at the first hipHostRegister (the first window); steady state from the 6th main kernel
regs=[r for r in api if r['Function']=='hipHostRegister']
t0=int(regs[0]['Start_Timestamp']) if regs else blk[0]
t1=blk[min(12,len(blk)-1)]
print('from the first window registration to the 13th main kernel: %.1f ms; first main kernel starts at %.1f ms' % ((t1-t0)/1e6,(blk[0]-t0)/1e6))
tot=collections.Counter(); n=collections.Counter(); big=[]
for r in api:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0-50_000_000<=s<=t1:
        tot[r['Function']]+=e-s; n[r['Function']]+=1
        if e-s>1_000_000: big.append(((s-t0)/1e6,(e-s)/1e6,r['Function'],r.get('Thread_Id','')))
for f,t in tot.most_common(12): print('   %-34s %5d calls %8.2f ms' % (f,n[f],t/1e6))
print('calls over 1 ms (start relative to the first registration, duration, call, thread):')
for b in big[:60]: print('   %8.2f  %7.2f  %-28s %s' % b)
PY
