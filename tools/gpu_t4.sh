#!/bin/bash
# GPU iteration for the 4-actor variant: its tests, then the whole GPU suite, then bench lines for TT and T4
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_t4_fused.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_t4.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_t4.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/pytest_t4.log; exit $rc; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/pytest_gpu.log; exit $rc; }
for v in TT T4; do
  for n in 16384 8192; do
    timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --variant $v --num-envs $n > gpurun_out/bench_$v_$n.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
    python - "$v" "$n" gpurun_out/bench_$v_$n.json <<'PY'
import json, sys
d=json.load(open(sys.argv[3]))
print("%s n=%s value %.1f M env-steps/s  kernel %.2f us  frac %.4f" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["roofline"]["avg_kernel_us"], d["roofline"]["frac"]))
PY
  done
done
