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
ORIG_LD_PRELOAD="${LD_PRELOAD:-}"
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
  echo "## ThreadSanitizer: build_bvh + flatten_scene on 8 threads (tools/tsan/drv.cpp; host sources only, device entry points stubbed)"
  ( export LD_PRELOAD="$ORIG_LD_PRELOAD"; unset ASAN_OPTIONS  # without the ASan runtime, whatever else was preloaded stays
    C=rust-raytracer_amd/csrc
    /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=thread -ffp-contract=off -DASSETS_DIR="\"$PWD/assets\"" -I$C tools/tsan/drv.cpp tools/tsan/stubs.cpp \
        $C/bvh_build.cpp $C/flatten.cpp $C/host_scene.cpp $C/host_io.cpp $C/scene_file.cpp -lz -lpthread -o /tmp/rtx_tsan_drv 2>&1 | grep -E "error" ; \
    TSAN_OPTIONS=halt_on_error=1 /tmp/rtx_tsan_drv 2>&1 | tail -8 && echo "tsan: no report" )
} > $LOG 2>&1 || { cat $LOG; exit 1; }
cat $LOG
