#!/bin/bash
# policy MLP: parity tests, then the forward alone (eager, per tile choice) at the two rollout sizes
timeout -k 10 300 python -m pytest tests/test_policy_mlp.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 || exit 1
for tile in ${TILES:-128 384 512 0}; do
PPENV_MLP_TILE=$tile timeout -k 10 300 python tools/gpu_mlp_layers.py 4096 313 || exit 1
PPENV_MLP_TILE=$tile timeout -k 10 300 python tools/gpu_mlp_layers.py 16384 80 || exit 1
done
