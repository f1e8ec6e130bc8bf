#!/bin/bash
# profiling: how the state stores leave the CU (PP_STORE_MODE 0 plain / 1 non-temporal / 2 system scope)
set -o pipefail
mkdir -p gpurun_out/sm
FL="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -disable-vector-combine -fno-signed-zeros -ffinite-math-only -fPIC -shared"
for m in 0 1 2; do
  hipcc $FL -DPP_STORE_MODE=$m -o gpurun_out/sm/lib$m.so isaacgym_amd/csrc/ppenv.hip isaacgym_amd/csrc/ppenv_ta.hip || exit 1
  for n in 16384 65536; do
    PPENV_LIB=$PWD/gpurun_out/sm/lib$m.so timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --num-envs $n > gpurun_out/sm/b.json 2> gpurun_out/sm/b.err || { tail -5 gpurun_out/sm/b.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/sm/b.json')); print('store mode $m  N=$n  kernel %.2f us  step %.2f us' % (d['roofline']['avg_kernel_us'], d['ms_per_step']*1e3))"
  done
done
