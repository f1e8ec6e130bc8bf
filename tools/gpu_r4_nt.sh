#!/bin/bash
# plain vs non-temporal activation stores in the policy layers' epilogue: per-layer times and the rollout step
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
for lib in default nt; do
  [ $lib = nt ] && export PPENV_LIB=$PWD/build_variants/ppnt/libppenv.so
  echo "== stores: $lib" | tee -a gpurun_out/mlp_nt.txt
  timeout -k 10 300 python tools/gpu_mlp_layers.py 4096 313 2>&1 | grep -v "amdgpu.ids\|fused" | cut -c1-72 | tee -a gpurun_out/mlp_nt.txt || exit 1
  timeout -k 10 300 python tools/rollout_bench.py --variant TA --num-envs 4096 --policy native > gpurun_out/rollout_ab.json 2> gpurun_out/rollout.err || { tail -20 gpurun_out/rollout.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/rollout_ab.json')); print('us_per_rollout_step %.1f  forward eager %.1f  env %.1f' % (d['us_per_rollout_step'], d['us_policy_forward_eager'], d['us_env_step_eager']))" | tee -a gpurun_out/mlp_nt.txt
done
