#!/bin/bash
# 27-dof evidence: bench line at BASELINE config 5's per-GPU size + rocprofv3 kernel trace of the same command
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python bench.py --variant TA --num-envs 4096 --steps 2000 --warmup 200 > gpurun_out/bench_TA.json 2> gpurun_out/bench_TA.err || { tail -20 gpurun_out/bench_TA.err; exit 1; }
cat gpurun_out/bench_TA.json
rm -rf gpurun_out/prof_TA
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_TA -- python bench.py --variant TA --num-envs 4096 --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/bench_TA_prof.json 2> gpurun_out/prof_TA.err || { tail -20 gpurun_out/prof_TA.err; exit 1; }
for f in $(find gpurun_out/prof_TA -name "*kernel_stats.csv"); do head -4 $f | cut -c1-300; done
