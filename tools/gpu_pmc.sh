#!/bin/bash
# usage (on the GPU box): tools/gpu_pmc.sh <config> <out-name> [counter groups...]   -- one rocprofv3 --pmc pass per group over bench.py --config <config> --steps 1,
# per-kernel sums printed and saved to gpurun_out/r04/pmc_<out-name>.txt.  Groups: sq (issue / wait cycles), grbm (clock), fetch, write, f64, f32
R=$GRAFT_REPO_ROOT; cfg=$1; name=$2; shift 2
mkdir -p $R/gpurun_out/r04
cd /tmp; export TMPDIR=/tmp
for g in "$@"; do
  case $g in
    sq) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA";;
    grbm) C="GRBM_GUI_ACTIVE";;
    fetch) C="FETCH_SIZE";;
    write) C="WRITE_SIZE";;
    f64) C="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64";;
    f32) C="SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_MOPS_F32";;
    lds) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR";;
    lds2) C="SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM";;
    mfma) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16";;
  esac
  rm -rf /tmp/pmc_$g
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$g -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-extra > /tmp/pmc_$g.log 2>&1
  python - <<PY | tee -a $R/gpurun_out/r04/pmc_$name.txt
import csv, glob, collections
f = glob.glob('/tmp/pmc_$g/*/*counter_collection.csv')
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.Counter(); seen = set()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if not k.startswith('k_'): continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); nd[k] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print('$cfg $g %-28s disp %3d  ' % (k[:28], nd[k]) + '  '.join('%s=%.4g' % (c, v) for c, v in sorted(acc[k].items())))
PY
done
