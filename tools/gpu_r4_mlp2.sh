#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_policy_mlp.py tests/test_policy_backward.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_mlp.log 2>&1 || { tail -30 gpurun_out/pytest_mlp.log; exit 1; }
tail -3 gpurun_out/pytest_mlp.log
timeout -k 10 300 python tools/gpu_mlp_layers.py 4096 313 2>&1 | grep -v "amdgpu.ids\|fused" | cut -c1-72 | tee gpurun_out/mlp_layers_4096.txt || exit 1
timeout -k 10 300 python tools/rollout_bench.py --variant TA --num-envs 4096 --policy native > gpurun_out/rollout_TA_native.json 2> gpurun_out/rollout.err || { tail -20 gpurun_out/rollout.err; exit 1; }
cut -c1-700 gpurun_out/rollout_TA_native.json
