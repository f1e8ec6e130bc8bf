#!/bin/bash
# Ablation timing (profiling only): full kernel vs pieces compiled out.
set -o pipefail
mkdir -p gpurun_out/ablate
FL="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -disable-vector-combine -fno-signed-zeros -ffinite-math-only -fPIC -shared"
for a in 0 1 2 3; do
  hipcc $FL -DPP_ABLATE=$a -o gpurun_out/ablate/lib$a.so isaacgym_amd/csrc/ppenv.hip || exit 1
  for n in 16384 65536; do
    PPENV_LIB=$PWD/gpurun_out/ablate/lib$a.so timeout -k 10 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --num-envs $n > gpurun_out/ablate/b$a.json 2> gpurun_out/ablate/b$a.err || { tail -5 gpurun_out/ablate/b$a.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ablate/b$a.json')); print('ablate $a  N=$n  kernel %.2f us' % d['roofline']['avg_kernel_us'])"
  done
done
