#!/bin/bash
# AddressSanitizer + UBSan run of the host-only L D L^H factorisation (CPU build: GPU sanitizers are not available on this pool):
# random real / Hermitian indefinite matrices with zero diagonal entries, fronts large enough for the threaded update.
# usage: tools/ldlt_sanitize.sh   (here or on any box; needs only the ROCm clang)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${TMPDIR:-/tmp}/rlh_ldlt_asan; mkdir -p $O
${HIPCLANG:-/opt/rocm/lib/llvm/bin/clang++} -x c++ -D__HIP_PLATFORM_AMD__ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer \
  -I $R/include -I $R/raleigh_amd/csrc -I ${ROCM_PATH:-/opt/rocm}/include $R/raleigh_amd/csrc/ldlt_host.cpp $R/tools/ldlt_sanitize_main.cpp \
  -o $O/ldlt_asan -lpthread
$O/ldlt_asan
