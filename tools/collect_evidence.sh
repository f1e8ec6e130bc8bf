#!/bin/bash
# After `gpurun -- bash tools/gpu_evidence.sh`: copy the judged summaries from gpurun_out/ (scratch) into profiles/ under a tag.
#   bash tools/collect_evidence.sh r02_d
set -e
tag=$1; go=gpurun_out; out=profiles
python tools/collect_profiles.py $tag
cp $go/bench_TA.json $out/${tag}_TA_bench.json
cp "$(ls -t $go/prof_TA/*/*kernel_stats.csv | head -1)" $out/${tag}_TA_kernel_stats.csv
cp $go/pmc_ta/summary.csv $out/${tag}_TA_pmc_summary.csv
cp $go/pmc_ta/traffic.json $out/${tag}_TA_pmc_traffic.json
cp $go/ta_chain_stamps.txt $out/${tag}_TA_chain_stamps.txt
for v in TA TT; do for p in native torch; do cp $go/rollout_${v}_${p}.json $out/${tag}_rollout_${v}_${p}.json; done; done
grep -v amdgpu.ids $go/mlp_layers.txt > $out/${tag}_mlp_layers.txt
for v in T4 TN T3; do cp $go/bench_$v.json $out/${tag}_${v}_bench.json; done
grep -h "passed\|probe excluded" $go/pytest_gpu.log > $out/${tag}_pytest_gpu.txt
ls $out | grep $tag
for v in TA TT; do cp "$(ls -t $go/prof_rollout_$v/*/*kernel_stats.csv | head -1)" $out/${tag}_rollout_${v}_kernel_stats.csv; done
cp $go/mlp_pmc_16384_summary.csv $out/${tag}_mlp_pmc_16384_summary.csv; cp $go/mlp_pmc_4096_summary.csv $out/${tag}_mlp_pmc_4096_summary.csv
cp $go/mlp_race_screen.txt $out/${tag}_mlp_race_screen.txt
