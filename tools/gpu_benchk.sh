#!/bin/bash
# bench.py at step counts a driver might pass (short runs, counts that are not a multiple of the 32-step graph)
set -o pipefail
mkdir -p gpurun_out
for spec in "TT 16384 20 5" "TT 16384 100 10" "TT 16384 2000 200" "T4 8192 1000 100" "T4 8192 1000 100" "T4 8192 2000 200"; do
  set -- $spec
  timeout -k 10 300 python bench.py --steps $3 --warmup $4 --no-cpu-baseline --variant $1 --num-envs $2 > gpurun_out/bench_k.json 2> gpurun_out/bench_k.err || { tail -20 gpurun_out/bench_k.err; exit 1; }
  python - "$@" <<'PY'
import json, sys
d=json.load(open("gpurun_out/bench_k.json"))
print("%s n=%s steps=%s warmup=%s: value %.1f M env-steps/s  ms_per_step %.5f  kernel %.2f us" % (*sys.argv[1:5], d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_kernel_us"]))
PY
done
