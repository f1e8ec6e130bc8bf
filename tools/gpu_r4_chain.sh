#!/bin/bash
# the chained-layer launch: parity, then the rollout with layers 5-6, 3-6, 2-6 chained against the per-layer launches (same box)
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 200 python -m pytest tests/test_policy_mlp.py -m gpu -x -q -p no:cacheprovider -k "chained" > gpurun_out/pytest_chain.log 2>&1 || { tail -30 gpurun_out/pytest_chain.log; exit 1; }
tail -2 gpurun_out/pytest_chain.log
for rep in 1 2; do for ch in none 5,6 3,6 3,4; do
  if [ $ch = none ]; then unset PPENV_MLP_CHAIN; else export PPENV_MLP_CHAIN=$ch; fi
  timeout -k 10 200 python tools/gpu_rollout_pipeline.py 4096 1 2>&1 | grep "^N=" | sed "s/^/chain=$ch  /" | cut -c1-100 | tee -a gpurun_out/chain_ab.txt || exit 1
done; done
