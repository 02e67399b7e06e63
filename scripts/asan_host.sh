#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libsplitvae_hip.so's HOST code (plan building, geometry, argument checks, the tape
# recorder; device code is compiled as usual and never run here) and the CPU tests that drive it.  CPU box only: GPU ASan / XNACK
# runs are not available on this pool.   usage: bash scripts/asan_host.sh [out.txt]
set -e
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r03_asan_host.txt}
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SV_LIB_NAME=libsplitvae_asan.so SV_OBJ_TAG=_asan \
  SV_EXTRA_FLAGS="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -g" \
  python split_vae_amd/build.py > /dev/null
{
  echo "# host ASan+UBSan build (-Xarch_host -fsanitize=address,undefined), runtime $ASAN"
  echo "# LD_PRELOAD=<asan runtime> SV_LIB_NAME=libsplitvae_asan.so python -m pytest tests/test_abi.py tests/test_host_logic.py -q -m 'not gpu'"
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    SV_LIB_NAME=libsplitvae_asan.so python -m pytest tests/test_abi.py tests/test_host_logic.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -5
} | tee "$OUT"
