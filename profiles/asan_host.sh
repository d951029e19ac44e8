#!/bin/bash
# usage (dev container, CPU only): profiles/asan_host.sh
# The host C library built with -fsanitize=address,undefined, and the CPU tests that drive it (tokenisers, parallel fill,
# formats, slot order, report printers, interop with the reference binary) run against that build.  The sanitizer runtime
# has to be loaded before the interpreter's allocator: LD_PRELOAD.  Leak checking is off (CPython itself "leaks").
set -e
cd "$(dirname "$0")/.."
make -C public_kssd_amd asan
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
export KSSD_HOST_LIB=$PWD/build/asan/libkssd_host.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
LD_PRELOAD="$ASAN $UBSAN" python -m pytest tests/test_host_fill.py tests/test_print_pairs.py tests/test_interop_ref.py tests/test_golden.py tests/test_number_formats.py tests/test_shuf_core.py tests/test_wide_tuples.py tests/test_allpairs_flow.py tests/test_inflate.py \
    -x -q -m "not gpu" 2>&1 | tail -15
