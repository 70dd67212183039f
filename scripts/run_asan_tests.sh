#!/bin/bash
# The CPU-side C-ABI tests under AddressSanitizer + UBSan (host code of capi.hip / mgpu.hip; SURVEY.md §5 aux: "ASan/UBSan build of the host wrappers").
# No GPU needed: tests/test_mgpu_mock.py drives vqhip_rowtile / vqhip_comm_* / vqhip_exchange_blur_halos / vqhip_composite_tiles with worlds of 2-4
# through the shared-memory RCCL stand-in (pitched tiles included), tests/test_abi.py the symbol table and error paths.
set -e
cd "$(dirname "$0")/.."
make -s -C vqengine_amd/csrc asan
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  VQHIP_LIBRARY_PATH=$PWD/vqengine_amd/lib/libvqhip_asan.so python -m pytest tests/test_mgpu_mock.py tests/test_abi.py -q "$@"
