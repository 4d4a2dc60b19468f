#!/bin/bash
# PMC passes on the fixed-effects kernels (development aid): where do the waves of k_glm_fast wait?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/glm_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/pmc_sq -- python $R/tools/gpu_probe_glm.py > $O/a.log 2>&1
rocprofv3 --pmc SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQ_BUSY_CYCLES SQ_INST_CYCLES_SMEM --kernel-trace --output-format csv -d $O/pmc_sqc -- python $R/tools/gpu_probe_glm.py > $O/b.log 2>&1
python - <<PY
import csv,glob,collections
for d in ("$O/pmc_sq","$O/pmc_sqc"):
    f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not f: print("no csv in",d); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0][-40:]
        if "k_glm_fast" in k or "k_firth" in k or "k_glm_final" in k:
            agg[(k,r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
    seen=set()
    for (k,dsp),c in agg.items():
        if k in seen: continue
        seen.add(k); print(k, dict(c))
PY
