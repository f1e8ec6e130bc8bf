#!/bin/bash
# quick GPU iteration: parity tests + bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && { tail -40 gpurun_out/pytest_gpu.log; exit $rc; }
timeout -k 10 600 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/bench_quick.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_quick.json"))
print("value %.1f M env-steps/s  kernel %.2f us  frac %.4f" % (d["value"]/1e6, d["roofline"]["avg_kernel_us"], d["roofline"]["frac"]))
PY
