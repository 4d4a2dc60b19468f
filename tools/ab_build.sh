#!/bin/bash
# A/B builds of one translation unit: tools/ab_build.sh <name> <file.hip> "<extra flags>"  ->  pyseer_amd/ab/libseerhip_<name>.so
# (the other objects are the current in-tree ones; SEERHIP_LIB=pyseer_amd/ab/libseerhip_<name>.so selects the build at run time)
set -e
cd "$(dirname "$0")/../pyseer_amd/csrc"
name=$1; src=$2; shift 2
base=${src%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-function "$@" -c $src -o ../ab/${base}_$name.o
objs=""
if [ "$base" = "firth_fast" ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-function "$@" -DFF_F32_TU -fno-slp-vectorize -c $src -o ../ab/firth_fast32_$name.o & fi
for o in api.o lmm_kernels.o glm_kernels.o glm_firth_v1.o glm_ols.o glm_lineage.o firth_rounds.o firth_fast.o firth_fast32.o firth_fast_w.o glm_wide.o bitdot_i8.o job_kernels.o dedup_kernels.o sim_kernels.o reader.o writer.o; do
  if [ "$o" = "$base.o" ]; then objs="$objs ../ab/${base}_$name.o"; elif [ "$base" = "firth_fast" ] && [ "$o" = "firth_fast32.o" ]; then objs="$objs ../ab/firth_fast32_$name.o"; else objs="$objs $o"; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/libseerhip_$name.so $objs -lz -lpthread
echo built pyseer_amd/ab/libseerhip_$name.so
