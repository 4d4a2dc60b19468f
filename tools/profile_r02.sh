#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun).  For every bench configuration:
#   1. the bench line itself                                            -> bench_<cfg>.json
#   2. rocprofv3 --kernel-trace --stats of the same command             -> stats_<cfg>/  (per-kernel average durations)
#   3. PMC passes in their own runs (one counter group per run): fp64/fp32 VALU instruction counts, MFMA ops, FETCH_SIZE, WRITE_SIZE,
#      SQ busy / wait cycles, GRBM_GUI_ACTIVE                            -> pmc_<cfg>_<group>/
# tools/summarize_r02.py then collapses them into profiles/r02/.  C3's PMC passes use tools/gpu_probe_lmm.py: torch.linalg.eigh
# (rocSOLVER) segfaults under counter collection, and the probe has the same kernels at N = 5000 without an eigensolver.
# NOTE: gpurun MERGES the box's gpurun_out/ into the local one: clear the local gpurun_out/r02 before a new run, or summarise on the box only
# (the script does: $O/summary)
# C3's counter passes run the probe at L = 4 limbs, what the automatic choice gives on the bench's kinship (LMM_L overrides)., otherwise tools/summarize_r02.py adds up the counter files of two runs.  PMC=0 skips the counter passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CFGS=${CFGS:-"C3 C2 C2N5000 C4"}
for c in $CFGS; do
  python $R/bench.py --config $c --steps 10 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python $R/bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-extra --no-parity > $O/stats_$c.json 2> $O/stats_$c.err
done
pmc() {  # cfg group counters...
  local c=$1 g=$2; shift 2
  if [ "$c" = "C3" ]; then
    L=${LMM_L:-4} V=262144 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${c}_$g -- python $R/tools/gpu_probe_lmm.py > $O/pmc_${c}_$g.log 2>&1
  else
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${c}_$g -- python $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-parity > $O/pmc_${c}_$g.log 2>&1
  fi
}
for c in $CFGS; do
  [ "${PMC:-1}" = "0" ] && continue
  pmc $c fetch FETCH_SIZE
  pmc $c write WRITE_SIZE
  pmc $c grbm GRBM_GUI_ACTIVE
  pmc $c sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
  if [ "$c" != "C3" ]; then
    pmc $c f64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64
    pmc $c f32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_MOPS_F32
  else
    pmc $c tcc TCC_HIT_sum TCC_MISS_sum
    pmc $c lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_LDS
  fi
done
python $R/tools/summarize_r02.py $O $O/summary
ls $O/summary
