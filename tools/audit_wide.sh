#!/bin/bash
# Build audit of k_lmm_quadform_i8w (cdna_hip_programming.md 5.7 item 4): the accumulators and the LDS reads of that kernel are inline assembly the
# compiler does not model, so after every edit: no scratch, no compiler-generated v_accvgpr_*, and the list of compiler instructions inside the stage
# loop (everything outside ;;#ASMSTART / ;;#ASMEND) printed for inspection.
cd "$(dirname "$0")/../pyseer_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o /tmp/lmm_w.s lmm_kernels.hip 2>/dev/null
K=${1:-_Z18k_lmm_quadform_i8wILi0ELi0ELi0EEvPKaPKmliiiPdPKii}
awk "/^$K:/,/\\.end_amdhsa_kernel/" /tmp/lmm_w.s > /tmp/w0.s
echo "lines $(wc -l < /tmp/w0.s)  mfma $(grep -c v_mfma /tmp/w0.s)  scratch $(grep -c scratch_ /tmp/w0.s)"
grep -E "next_free_vgpr|accum_offset|private_segment_fixed|next_free_sgpr" /tmp/w0.s
echo "compiler v_accvgpr outside asm: $(awk '/ASMSTART/{a=1} /ASMEND/{a=0} !a && /v_accvgpr/' /tmp/w0.s | wc -l)"
echo "compiler waitcnt outside asm:"; awk '/ASMSTART/{a=1} /ASMEND/{a=0} !a && /s_waitcnt/' /tmp/w0.s | sort | uniq -c
