#!/bin/bash
# ThreadSanitizer over the host concurrency that only runs with a device behind it (round 5's additions: csrc/lanes_api.inc -- the lanes'
# worker threads, csrc/job_api.inc -- the job stream's ring, copy stream and collect, csrc/host_pool.h -- the process-wide CPU pool and the
# janitor thread; all host code of api.hip / reader.cpp / writer.cpp).  tools/sanitize_host.sh covers what runs without a GPU.
#   tools/sanitize_gpu.sh build   (build container: hipcc cross-compiles; the instrumented library travels with the snapshot)
#   tools/sanitize_gpu.sh run     (GPU box)  -> gpurun_out/r06/sanitizers_gpu.txt
# tests/test_job_gpu.py drives all of it without torch (torch's lazy dlopen of its nvrtc stub fails under the preloaded runtime): LMM and
# fixed-effects jobs (the latter compute their blocks on the context's LANES), registered-window DMA, two contexts with two jobs side by side.
# Host code is instrumented by clang (-fsanitize=thread applies to the host side of a HIP translation unit only); the HIP runtime is not,
# so reports whose every frame is inside libamdhip64 / libhsa-runtime64 are suppressed (tools/tsan_gpu.supp) -- a report with a frame of
# this library in it is not.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C="$R/pyseer_amd/csrc"; AB="$R/pyseer_amd/ab"; mkdir -p "$AB"
LLVM=/opt/rocm/lib/llvm
RT=$(ls $LLVM/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
if [ "$1" = build ]; then
  F="-O1 -g -std=c++17 -fPIC -fsanitize=thread -fno-omit-frame-pointer"
  ( cd "$C"; /opt/rocm/bin/hipcc $F --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-unused-function -c api.hip -o "$AB/api_tsan.o" 2>/dev/null )
  $LLVM/bin/clang++ $F -pthread -c "$C/reader.cpp" -o "$AB/reader_tsan.o"
  $LLVM/bin/clang++ $F -pthread -c "$C/writer.cpp" -o "$AB/writer_tsan.o"
  objs=""
  for o in lmm_kernels.o glm_kernels.o glm_firth_v1.o glm_ols.o glm_lineage.o firth_rounds.o firth_fast.o firth_fast32.o firth_fast_w.o glm_wide.o bitdot_i8.o job_kernels.o dedup_kernels.o sim_kernels.o; do objs="$objs $C/$o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -shared-libsan -o "$AB/libseerhip_gtsan.so" "$AB/api_tsan.o" $objs "$AB/reader_tsan.o" "$AB/writer_tsan.o" -lz -lpthread
  rm -f "$AB/api_tsan.o" "$AB/reader_tsan.o" "$AB/writer_tsan.o"
  echo "built $AB/libseerhip_gtsan.so (runtime: $RT)"
  exit 0
fi
OUT="$R/gpurun_out/r06"; mkdir -p "$OUT"
cd "$R"
export LD_PRELOAD="$RT" SEERHIP_LIB="$AB/libseerhip_gtsan.so"
export TSAN_OPTIONS="report_signal_unsafe=0 suppressions=$R/tools/tsan_gpu.supp history_size=3 exitcode=0 print_suppressions=1 log_path=/tmp/tsan_gpu"
rm -f /tmp/tsan_gpu.*
timeout 1500 python -m pytest -q -m gpu -p no:cacheprovider tests/test_job_gpu.py \
  tests/test_sink_cpu.py > /tmp/sanitize_gpu.log 2>&1 || true
unset LD_PRELOAD
n=$(cat /tmp/tsan_gpu.* 2>/dev/null | grep -c "WARNING: ThreadSanitizer" || true)
ours=$(cat /tmp/tsan_gpu.* 2>/dev/null | grep -c "libseerhip_gtsan" || true)
{
  echo "tools/sanitize_gpu.sh run (MI355X box; host code of api.hip incl. lanes_api.inc / job_api.inc / host_pool.h, reader.cpp, writer.cpp under ThreadSanitizer):"
  echo "pytest: $(tail -1 /tmp/sanitize_gpu.log)"
  echo "ThreadSanitizer reports not suppressed: $n; lines naming this library in them: $ours"
  cat /tmp/tsan_gpu.* 2>/dev/null | head -150
  grep -h -A12 "Suppressions (pid" /tmp/sanitize_gpu.log | head -40
} > "$OUT/sanitizers_gpu.txt"
grep -E "^E  " /tmp/sanitize_gpu.log | head -30; tail -3 /tmp/sanitize_gpu.log; head -5 "$OUT/sanitizers_gpu.txt"
