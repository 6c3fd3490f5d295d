#!/bin/bash
# The library's host code (readers, database, result writers, host routines) under ASAN + UBSAN on the CPU box: builds
# build/asan/libstarphase_hip_asan.so (make -C pb-starphase_amd/csrc asan) and runs the CPU tests that drive that code through the C ABI.
set -u
cd "$(dirname "$0")/../.."
make -s -j4 -C pb-starphase_amd/csrc asan || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export SP_LIB_PATH=$PWD/build/asan/libstarphase_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$RT python -m pytest tests/test_io.py tests/test_database.py tests/test_host_functions.py tests/test_cyp_db.py tests/test_debug_files.py tests/test_abi.py -x -q -m "not gpu" -p no:cacheprovider "$@"
