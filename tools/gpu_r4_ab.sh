#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for ring in 0 1; do
  PPENV_MLP_RING=$ring timeout -k 10 300 python tools/rollout_bench.py --variant TA --num-envs 4096 --policy native > gpurun_out/rollout_ab.json 2> gpurun_out/rollout.err || { tail -20 gpurun_out/rollout.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/rollout_ab.json')); print('ring=$ring rep=$rep  us_per_rollout_step %.1f  forward eager %.1f  env %.1f' % (d['us_per_rollout_step'], d['us_policy_forward_eager'], d['us_env_step_eager']))" | tee -a gpurun_out/rollout_ab.txt
done; done
