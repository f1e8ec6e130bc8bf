#!/bin/bash
# the driver's short run (--steps 20 --warmup 5): the 20 steps as one native burst (ppenv_step_sequence, default) / one 20-step graph replay / 20 Python calls
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "step_sequence" 2>&1 | tail -1
for rep in 1 2 3; do for mode in burst graph eager; do
  flag=""; [ $mode = eager ] && flag="--no-graph"
  burst=1; [ $mode = graph ] && burst=0
  PPENV_BENCH_BURST=$burst timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs $flag > gpurun_out/bench_small.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/bench_small.json')); print('$mode rep $rep: value %.3f G env-steps/s  ms_per_step %.4f' % (d['value']/1e9, d['ms_per_step']))" | tee -a gpurun_out/smallk.txt
done; done
PPENV_BENCH_BURST=1 timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2000 steps: value %.3f G  ms_per_step %.5f' % (d['value']/1e9, d['ms_per_step']))" | tee -a gpurun_out/smallk.txt
