#!/bin/bash
# same-box A/B: HEAD's library without kernarg preload (build_variants/tt_base) against the working tree's (leading pointer arguments preloaded into SGPRs)
set -o pipefail
[ -f build_variants/tt_base/libppenv.so ] || { echo "build_variants/tt_base/libppenv.so is missing: build the baseline first, in the container — git stash; python -c \"from isaacgym_amd import _lib; _lib.build(out='build_variants/tt_base/libppenv.so', force=True)\"; git stash pop"; exit 2; }
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_t4_fused.py tests/test_pins_golden.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_pre.log 2>&1 || { tail -30 gpurun_out/pytest_pre.log; exit 1; }
tail -1 gpurun_out/pytest_pre.log
for rep in 1 2 3; do for lib in base new; do
  if [ $lib = base ]; then export PPENV_LIB=$PWD/build_variants/tt_base/libppenv.so; else unset PPENV_LIB; fi
  for v in TT:16384 T4:8192; do
    timeout -k 10 300 python bench.py --variant ${v%%:*} --num-envs ${v##*:} --steps 1024 --warmup 128 --no-cpu-baseline --no-configs > gpurun_out/bench_pre.json 2>/dev/null || exit 1
    python -c "
import json; d=json.load(open('gpurun_out/bench_pre.json')); print('$lib rep $rep  ${v%%:*} ${v##*:}: %.3f us' % d['roofline']['avg_kernel_us'])" | tee -a gpurun_out/preload_ab.txt
  done
done; done
