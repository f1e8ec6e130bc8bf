#!/bin/bash
# Round 4, first call: tests / smoke / bench / trace / PMC of the restored build, then the round's measurement-driven items
# (T4 PMC, DR timings of the 7-dof and the 27-dof kernels, the pipelined rollout, the multi-stream tool).
set -o pipefail
bash tools/gpu_round.sh || exit 1
bash tools/gpu_t4_pmc.sh > gpurun_out/t4_pmc.txt 2>&1 || { tail -20 gpurun_out/t4_pmc.txt; exit 1; }
tail -3 gpurun_out/t4_pmc.txt
timeout -k 10 200 python tools/gpu_ta_dr_time.py 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ta_dr_time.txt || exit 1
timeout -k 10 200 python tools/gpu_dr_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dr_time.txt || exit 1
timeout -k 10 300 python tools/gpu_rollout_pipeline.py 4096 1,2,4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rollout_pipeline.txt || exit 1
timeout -k 10 300 python tools/gpu_multistream.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/multistream.txt || exit 1
