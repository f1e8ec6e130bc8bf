#!/bin/bash
# experiment: non-temporal obs stores
set -o pipefail
mkdir -p gpurun_out/ablate
FL="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -disable-vector-combine -fno-signed-zeros -ffinite-math-only -fPIC -shared"
hipcc $FL -DPP_OBS_NT=1 -o gpurun_out/ablate/libnt.so isaacgym_amd/csrc/ppenv.hip isaacgym_amd/csrc/ppenv_ta.hip || exit 1
for rep in 1 2; do
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('plain stores  kernel %.2f us' % d['roofline']['avg_kernel_us'])"
PPENV_LIB=$PWD/gpurun_out/ablate/libnt.so python bench.py --steps 2000 --warmup 200 --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin); print('nt obs stores kernel %.2f us' % d['roofline']['avg_kernel_us'])"
done
