#!/bin/bash
# The whole evidence set of a build in one call (≈4 GPU-minutes): tests / smoke / bench / traces / PMC for the headline kernel, then the
# 27-dof step (bench + trace + PMC + in-kernel timeline) and the rollout loop with the native policy forward.
set -o pipefail
bash tools/gpu_round.sh || exit 1
bash tools/gpu_ta_prof.sh || exit 1
bash tools/gpu_ta_pmc.sh > gpurun_out/ta_pmc.txt 2>&1 || { tail -20 gpurun_out/ta_pmc.txt; exit 1; }
tail -4 gpurun_out/ta_pmc.txt
python tools/gpu_ta_chain_stamps.py 4096 > gpurun_out/ta_chain_stamps.txt 2>&1 || { tail -20 gpurun_out/ta_chain_stamps.txt; exit 1; }
tail -3 gpurun_out/ta_chain_stamps.txt
for pol in native torch; do for v in TA TT; do n=4096; [ $v = TT ] && n=16384
  timeout -k 10 300 python tools/rollout_bench.py --variant $v --num-envs $n --policy $pol > gpurun_out/rollout_${v}_${pol}.json 2> gpurun_out/rollout.err || { tail -20 gpurun_out/rollout.err; exit 1; }
  cut -c1-400 gpurun_out/rollout_${v}_${pol}.json
done; done
python tools/gpu_mlp_layers.py 4096 313 > gpurun_out/mlp_layers.txt 2>&1; python tools/gpu_mlp_layers.py 16384 80 >> gpurun_out/mlp_layers.txt 2>&1; grep -v amdgpu.ids gpurun_out/mlp_layers.txt
for v in T4:8192 TN:16384 T3:16384; do timeout -k 10 300 python bench.py --variant ${v%%:*} --num-envs ${v##*:} --steps 1024 --warmup 128 --no-cpu-baseline > gpurun_out/bench_${v%%:*}.json 2>/dev/null && python -c "
import json,sys; d=json.load(open('gpurun_out/bench_${v%%:*}.json')); print('${v%%:*}', d['config']['num_envs_per_gpu'], '%.2f us  %.0f M env-steps/s  frac %.4f' % (d['roofline']['avg_kernel_us'], d['value']/1e6, d['roofline']['frac']))"; done
bash tools/gpu_rollout_prof.sh || exit 1
MLP_M=16384 MLP_K=80 bash tools/gpu_mlp_pmc.sh > gpurun_out/mlp_pmc_16384.txt 2>&1 && cp gpurun_out/pmc_mlp/summary.csv gpurun_out/mlp_pmc_16384_summary.csv && tail -8 gpurun_out/mlp_pmc_16384.txt
MLP_M=4096 MLP_K=313 bash tools/gpu_mlp_pmc.sh > gpurun_out/mlp_pmc_4096.txt 2>&1 && cp gpurun_out/pmc_mlp/summary.csv gpurun_out/mlp_pmc_4096_summary.csv && tail -8 gpurun_out/mlp_pmc_4096.txt
timeout -k 10 500 python tools/gpu_mlp_race_screen.py 200 2>&1 | grep -v amdgpu.ids > gpurun_out/mlp_race_screen.txt; tail -1 gpurun_out/mlp_race_screen.txt
