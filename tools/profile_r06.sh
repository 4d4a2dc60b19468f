#!/bin/bash
# Round-6 profiling recipe (run on the GPU box through gpurun).  For every bench configuration:
#   1. the bench line itself (C3 = the default invocation, i.e. what the driver runs, `extra` included)   -> bench_<cfg>.json
#   2. rocprofv3 --kernel-trace --stats of the same command (without the CPU legs)                          -> stats_<cfg>/
#   3. PMC passes in their own runs, one counter group per run (never combined with the trace domains gpurun refuses): FETCH_SIZE,
#      WRITE_SIZE, GRBM_GUI_ACTIVE, SQ issue / wait cycles + SQ_INSTS_VALU, fp64 / fp32 / f16 instruction counts (fixed effects), L2 hits
#      and LDS (C3).  SEERHIP_BENCH_CPU_EIGH=1: rocSOLVER's eigensolver segfaults under counter collection.
# tools/summarize_prof.py collapses them into gpurun_out/r06/profiles (copied to profiles/r06/).  CFGS / PMC=0 select a subset.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CFGS=${CFGS:-"C3 C2N5000 C4 C2"}
for c in $CFGS; do
  if [ "$c" = "C3" ]; then python $R/bench.py --steps 10 --warmup 2 > $O/bench_$c.json 2> $O/bench_$c.err
  else python $R/bench.py --config $c --steps 12 --warmup 6 > $O/bench_$c.json 2> $O/bench_$c.err; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python $R/bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline --no-extra --no-parity > $O/stats_$c.json 2> $O/stats_$c.err
  # fixed effects: the same steps on ONE lane (no two batches on the device at once): per-kernel durations that are the kernels' own
  [ "$c" != "C3" ] && SEERHIP_ROUTE=lanes=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1_$c -- python $R/bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline --no-extra --no-parity > $O/stats1_$c.json 2> $O/stats1_$c.err
done
pmc() {  # cfg group counters...
  local c=$1 g=$2; shift 2
  SEERHIP_BENCH_CPU_EIGH=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${c}_$g -- python $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-extra > $O/pmc_${c}_$g.log 2>&1
}
for c in $CFGS; do
  [ "${PMC:-1}" = "0" ] && continue
  pmc $c fetch FETCH_SIZE
  pmc $c write WRITE_SIZE
  if [ "$c" != "C3" ]; then
    pmc $c sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
    pmc $c f64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64
    pmc $c f32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_MOPS_F32
    pmc $c f16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_MFMA
  else
    pmc $c grbm GRBM_GUI_ACTIVE
    pmc $c sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
    pmc $c tcc TCC_HIT_sum TCC_MISS_sum
  fi
done
python $R/tools/summarize_prof.py $O $R/gpurun_out/r06/profiles
ls $R/gpurun_out/r06/profiles
