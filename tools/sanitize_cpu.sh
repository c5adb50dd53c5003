#!/bin/bash
# The CPU test suite against a build of the library whose HOST code carries AddressSanitizer + UndefinedBehaviorSanitizer
# (-Xarch_host: the gfx950 code objects are the product's; GPU sanitizers are not available on the pool).  Covers what the
# library does without a GPU: C/A codes, Doppler tables, host decision replay, manager / ring mirrors, bit and frame sync,
# argument checks and error paths of every entry the CPU tests reach.  Runs here (no GPU needed): a few minutes.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)")
[ -f "$RT/libclang_rt.asan-x86_64.so" ] || RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
GM_EXTRA_FLAGS="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -g" GM_LIB_SUFFIX=_asan \
    python3 "$R/gnss-sdr-rs_amd/build.py" > /dev/null
cd "$R"
LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 GM_LIB_PATH="$R/gnss-sdr-rs_amd/lib/libgnss_mi355x_asan.so" \
    python3 -m pytest tests -x -q -m "not gpu" "$@"
