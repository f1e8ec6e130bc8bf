#!/bin/bash
# Round 4: the 27-dof chain kernel after the early flush — parity tests that touch it, step time, rollout with / without the env writing the policy input
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_ta_physics.py tests/test_ta_golden.py tests/test_policy_mlp.py tests/test_urdf.py tests/test_isaacgymenvs_shim.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_ta.log 2>&1 || { tail -30 gpurun_out/pytest_ta.log; exit 1; }
tail -3 gpurun_out/pytest_ta.log
timeout -k 10 300 python bench.py --variant TA --num-envs 4096 --steps 1024 --warmup 128 --no-cpu-baseline --no-configs > gpurun_out/bench_TA.json 2>/dev/null || exit 1
python -c "
import json; d=json.load(open('gpurun_out/bench_TA.json')); print('TA 4096: %.2f us  %.1f M env-steps/s  frac %.4f' % (d['roofline']['avg_kernel_us'], d['value']/1e6, d['roofline']['frac']))" | tee gpurun_out/ta_time.txt
timeout -k 10 200 python tools/gpu_ta_chain_stamps.py 4096 > gpurun_out/ta_chain_stamps.txt 2>&1 || { tail -20 gpurun_out/ta_chain_stamps.txt; exit 1; }
grep "B1 leave\|B2\|B3\|^end\|inputs staged\|span" gpurun_out/ta_chain_stamps.txt
PIPE_ATTACH=0 timeout -k 10 200 python tools/gpu_rollout_pipeline.py 4096 1 2>&1 | grep "^N=" | tee -a gpurun_out/ta_time.txt
PIPE_ATTACH=1 timeout -k 10 200 python tools/gpu_rollout_pipeline.py 4096 1 2>&1 | grep "^N=" | tee -a gpurun_out/ta_time.txt
