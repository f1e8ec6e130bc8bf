#!/bin/bash
# Round 4 evidence set of the final build: tools/gpu_evidence.sh (tests / smoke / bench / traces / PMC / rollouts / policy layers / race screen)
# + the 4-actor counter passes + the randomisation timings.
set -o pipefail
bash tools/gpu_evidence.sh || exit 1
bash tools/gpu_t4_pmc.sh > gpurun_out/t4_pmc.txt 2>&1 || { tail -20 gpurun_out/t4_pmc.txt; exit 1; }
tail -2 gpurun_out/t4_pmc.txt
timeout -k 10 200 python tools/gpu_ta_dr_time.py 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ta_dr_time.txt || exit 1
timeout -k 10 200 python tools/gpu_dr_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dr_time.txt || exit 1
# gpurun copies back at most 64 MiB: summarise the headline counter passes here and drop every raw per-dispatch csv (they are most of the volume)
PROFILES_OUT=gpurun_out/summary python tools/collect_profiles.py r04_c > /dev/null || exit 1
find gpurun_out -name "*counter_collection.csv" -delete; find gpurun_out -name "*kernel_trace.csv" -delete; rm -f gpurun_out/libppenv_*.so
du -sh gpurun_out
