#!/bin/bash
# CPU box: builds the backend's host code and the oracle with AddressSanitizer + UBSan, runs the CPU test files
# that exercise them (host logic, oracle known answers, distributed packing) and the parser fuzz under them,
# and writes the log to profiles/<round>_sanitizers.txt.  (No GPU sanitizer runs: not available on this pool.)
set -e
cd "$(dirname "$0")/.."
ROUND=${1:-r02}
make -C rust-raytracer_amd/csrc asan -j8 > /tmp/asan_build.log 2>&1 || { tail -20 /tmp/asan_build.log; exit 1; }
make -C oracle asan >> /tmp/asan_build.log 2>&1 || { tail -20 /tmp/asan_build.log; exit 1; }
RTL=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD="$RTL${LD_PRELOAD:+:$LD_PRELOAD}"  # the ASan runtime first, whatever was preloaded stays
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export RTX_LIB=$PWD/rust-raytracer_amd/librtx_hip_asan.so ORC_LIB=$PWD/oracle/liboracle_asan.so
LOG=profiles/${ROUND}_sanitizers.txt
{
  echo "# host code of librtx_hip.so + oracle under -fsanitize=address,undefined (clang $(/opt/rocm/lib/llvm/bin/clang++ --version | head -1))"
  echo "## pytest tests/test_host.py tests/test_oracle_kat.py"
  python -m pytest tests/test_host.py tests/test_oracle_kat.py -q -x -p no:cacheprovider 2>&1 | tail -4
  echo "## tools/fuzz_parsers.py ${FUZZ_CASES:-300}"
  python tools/fuzz_parsers.py ${FUZZ_CASES:-300} 2>&1 | tail -12
} > $LOG 2>&1 || { cat $LOG; exit 1; }
cat $LOG
