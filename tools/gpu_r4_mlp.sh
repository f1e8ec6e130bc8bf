#!/bin/bash
# Round 4: policy-forward kernels — parity tests of the tiles, per-layer times (auto and forced tiles), in-kernel phases at the rollout's M = 4096.
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_policy_mlp.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_mlp.log 2>&1 || { tail -30 gpurun_out/pytest_mlp.log; exit 1; }
tail -3 gpurun_out/pytest_mlp.log
for t in ${MLP_TILES:-auto 518 519}; do
  if [ $t = auto ]; then unset PPENV_MLP_TILE; else export PPENV_MLP_TILE=$t; fi
  timeout -k 10 300 python tools/gpu_mlp_layers.py 4096 313 2>&1 | grep -v "amdgpu.ids\|fused" | cut -c1-72 | tee -a gpurun_out/mlp_layers_4096.txt || exit 1
done
unset PPENV_MLP_TILE
[ -n "$MLP_PHASES" ] && { timeout -k 10 300 python tools/gpu_mlp_phases.py 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mlp_phases_4096.txt || exit 1; }
exit 0
