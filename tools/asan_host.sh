#!/bin/bash
# The C++ host library (parsers, packer, table IO, BAM reader) under AddressSanitizer + UBSan on the CPU test-suite.
# CPU build only (GPU sanitizers are not available on this pool).  libstdc++ is preloaded beside libasan: in a Python process
# ASan otherwise finds no __cxa_throw to intercept and aborts at the first C++ exception.
set -e
cd "$(dirname "$0")/.."
D=${TMPDIR:-/tmp}/amplisolve_asan
mkdir -p $D && rm -f $D/asan.log* $D/ubsan.log*
g++ -O1 -g -fPIC -std=c++17 -ffp-contract=off -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o $D/libamplisolve_host.so $(ls amplisolve_amd/csrc/host/*.cpp | grep -v _main.cpp) -ldl -lz
AMPLISOLVE_HOST_LIB=$D/libamplisolve_host.so \
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" \
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:log_path=$D/asan.log UBSAN_OPTIONS=print_stacktrace=1:log_path=$D/ubsan.log \
python -m pytest tests/test_stream_ingest.py tests/test_host_logic.py tests/test_pileup_host.py tests/test_oracle_golden.py -x -q -m "not gpu" -p no:cacheprovider
if ls $D/asan.log* $D/ubsan.log* >/dev/null 2>&1; then echo "sanitizer reports:"; cat $D/asan.log* $D/ubsan.log* | head -80; exit 1; fi
echo "clean: no AddressSanitizer / UBSan report"
