#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; export V=262144
for sp in 1 0; do
  export SEERHIP_QF_SPLIT=$sp
  rm -rf /tmp/pf$sp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf$sp -- python $R/tools/gpu_probe_lmm.py > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pt$sp -- python $R/tools/gpu_probe_lmm.py > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
for d in ("/tmp/pf$sp","/tmp/pt$sp"):
    f=glob.glob(d+"/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "quadform" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("split=$sp",k,sum(v)/len(v))
PY
done
