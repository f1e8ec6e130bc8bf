#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side native code: the oracle (gcc) and the kernel-arithmetic host shim (g++).
# GPU sanitizers are not available on the pool; this covers the same per-env arithmetic on the host.
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
cp oracle/libppenv_oracle.so "$tmp/oracle.bak" 2>/dev/null || true
cp tests/csrc/libppenv_hostshim.so "$tmp/shim.bak" 2>/dev/null || true
restore() {
  [ -f "$tmp/oracle.bak" ] && cp "$tmp/oracle.bak" oracle/libppenv_oracle.so && touch oracle/libppenv_oracle.so
  [ -f "$tmp/shim.bak" ] && cp "$tmp/shim.bak" tests/csrc/libppenv_hostshim.so && touch tests/csrc/libppenv_hostshim.so
}
trap restore EXIT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
gcc $SAN -fPIC -fopenmp -ffp-contract=off -shared -o oracle/libppenv_oracle.so oracle/ppenv_oracle.c -lm
g++ $SAN -fPIC -shared -std=c++17 -ffp-contract=fast -fno-signed-zeros -ffinite-math-only -Wno-unknown-pragmas \
    -o tests/csrc/libppenv_hostshim.so tests/csrc/host_shim.cpp
touch oracle/libppenv_oracle.so tests/csrc/libppenv_hostshim.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
python -m pytest tests/test_oracle_golden.py tests/test_kernel_math_host.py tests/test_ta_golden.py tests/test_t4_golden.py tests/test_t4_fused.py tests/test_ta_physics.py \
    tests/test_host_logic.py -x -q -m "not gpu" -k "not layout and not exports"
