#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, bench, rocprofv3 kernel trace.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee gpurun_out/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -25 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc" | tee -a gpurun_out/progress.log
[ $rc -ne 0 ] && exit $rc
echo "== smoke" | tee -a gpurun_out/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log || exit 1
echo "== bench" | tee -a gpurun_out/progress.log
timeout -k 10 600 python bench.py --steps 2000 --warmup 200 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
echo "== rocprofv3 kernel trace" | tee -a gpurun_out/progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-configs > gpurun_out/bench_prof.json 2> gpurun_out/prof.err || { tail -20 gpurun_out/prof.err; exit 1; }
cat gpurun_out/bench_prof.json
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -12 $f; done
