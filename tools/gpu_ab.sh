#!/bin/bash
# A/B perf iteration: TT bench lines at several sizes, T4 at its config size (no parity tests)
set -o pipefail
mkdir -p gpurun_out
for spec in "TT 16384" "TT 65536" "T4 8192" ${EXTRA_SPECS}; do
  set -- $spec
  timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --variant $1 --num-envs $2 > gpurun_out/bench_ab.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
  python - "$1" "$2" <<'PY'
import json, sys
d=json.load(open("gpurun_out/bench_ab.json"))
print("%s n=%s value %.1f M env-steps/s  kernel %.2f us  frac %.4f" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["roofline"]["avg_kernel_us"], d["roofline"]["frac"]))
PY
done
