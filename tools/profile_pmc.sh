#!/bin/bash
# PMC passes on the LMM hot kernel (separate runs per counter group, as MI355X_MICROARCH.md prescribes).
# Uses tools/gpu_probe_lmm.py (no rocSOLVER in the process: torch.linalg.eigh segfaults under counter collection).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export V=262144
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/gpu_probe_lmm.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/gpu_probe_lmm.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -- python $R/tools/gpu_probe_lmm.py > $O/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python $R/tools/gpu_probe_lmm.py > $O/pmc_tcc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm -- python $R/tools/gpu_probe_lmm.py > $O/pmc_grbm.log 2>&1
find $O -name "*counter_collection.csv" | head; tail -3 $O/pmc_fetch.log
