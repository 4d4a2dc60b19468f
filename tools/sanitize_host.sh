#!/bin/bash
# Sanitizer builds of the HOST code of libseerhip (reader.cpp: producer / decoding / translating / parsing threads; writer.cpp: OpenMP
# formatter) linked with the current device objects, and the CPU tests of the reader and the sink under them (race and
# memory checking).  No GPU needed.  Usage: tools/sanitize_host.sh [asan|tsan|both]   -> /tmp/sanitize_<kind>.log, one summary line each.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C="$R/pyseer_amd/csrc"; AB="$R/pyseer_amd/ab"; mkdir -p "$AB"
objs=""
for o in api.o lmm_kernels.o glm_kernels.o glm_firth_v1.o glm_ols.o glm_lineage.o firth_rounds.o firth_fast.o firth_fast32.o firth_fast_w.o glm_wide.o bitdot_i8.o job_kernels.o dedup_kernels.o sim_kernels.o; do objs="$objs $C/$o"; done
gccdir=$(dirname "$(gcc -print-file-name=libasan.so)")
run() {  # kind flags preload tests...
  local kind=$1 flags=$2 pre=$3; shift 3
  g++ -O1 -g -std=c++17 -fPIC -fopenmp $flags -fno-omit-frame-pointer -c "$C/reader.cpp" -o "$AB/reader_$kind.o"
  g++ -O1 -g -std=c++17 -fPIC -fopenmp $flags -fno-omit-frame-pointer -c "$C/writer.cpp" -o "$AB/writer_$kind.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$AB/libseerhip_$kind.so" $objs "$AB/reader_$kind.o" "$AB/writer_$kind.o" -lz -lgomp $flags 2>/dev/null
  ( cd "$R"; LD_PRELOAD="$pre" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS="report_signal_unsafe=0" \
      SEERHIP_LIB="$AB/libseerhip_$kind.so" python -m pytest "$@" -q > /tmp/sanitize_$kind.log 2>&1 || true )
  echo "$kind: $(tail -1 /tmp/sanitize_$kind.log); reports: $(grep -c 'runtime error\|ERROR: AddressSanitizer\|WARNING: ThreadSanitizer' /tmp/sanitize_$kind.log)"
  rm -f "$AB/reader_$kind.o" "$AB/writer_$kind.o" "$AB/libseerhip_$kind.so"
}
what=${1:-both}
if [ "$what" = asan ] || [ "$what" = both ]; then
  run asan "-fsanitize=address,undefined" "$gccdir/libasan.so $gccdir/libubsan.so" tests/test_reader_cpu.py tests/test_sink_cpu.py
fi
if [ "$what" = tsan ] || [ "$what" = both ]; then
  run tsan "-fsanitize=thread" "$gccdir/libtsan.so" tests/test_reader_cpu.py -k "several_threads or container or corrupt or bounded or tokens or several_files"
fi
