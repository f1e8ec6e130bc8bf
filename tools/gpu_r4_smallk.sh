#!/bin/bash
# the driver's short run (--steps 20 --warmup 5): one 20-step graph replay against 20 eager launches
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for mode in graph eager; do
  flag=""; [ $mode = eager ] && flag="--no-graph"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs $flag > gpurun_out/bench_small.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/bench_small.json')); print('$mode rep $rep: value %.3f G env-steps/s  ms_per_step %.4f' % (d['value']/1e9, d['ms_per_step']))" | tee -a gpurun_out/smallk.txt
done; done
