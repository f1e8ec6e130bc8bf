#!/bin/bash
# Round 4 soaks on the final build: the new max-size test, then the step kernels against the oracle on seeds the suite does not use (plain and with
# domain randomisation), the 27-dof chain kernel, the backward race screen, the closed-loop rollout under the native policy.
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "largest_size or full_size" 2>&1 | tail -2 | tee gpurun_out/soaks.txt || exit 1
timeout -k 10 1000 python tools/gpu_soak.py 150 41 42 43 44 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/soaks.txt || exit 1
PPENV_SOAK_DR=1 timeout -k 10 600 python tools/gpu_soak.py 100 45 46 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/^/with domain randomisation: /' | tee -a gpurun_out/soaks.txt || exit 1
timeout -k 10 900 python tools/gpu_soak_ta_chain.py 2048 400 51 52 53 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/soaks.txt || exit 1
timeout -k 10 500 python tools/gpu_mlp_bwd_race_screen.py 150 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/soaks.txt || exit 1
timeout -k 10 500 python tools/gpu_rollout_soak.py TA 4096 20000 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/soaks.txt || exit 1
