#!/bin/bash
# Round 3 check on the GPU box (via gpurun): GPU parity tests, smoke, the bench line with every single-GPU config, kernel trace of the headline.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee gpurun_out/progress.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -40 gpurun_out/pytest_gpu.log; echo "pytest rc=$rc" | tee -a gpurun_out/progress.log
[ $rc -ne 0 ] && exit $rc
echo "== smoke" | tee -a gpurun_out/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log || exit 1
echo "== bench" | tee -a gpurun_out/progress.log
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print("headline %.3f G env-steps/s, kernel %.2f us, frac %.4f" % (d["value"] / 1e9, d["roofline"]["avg_kernel_us"], d["roofline"]["frac"]))
for r in d.get("configs", []):
    k = r.get("avg_kernel_us", r.get("us_per_rollout_step", r["roofline"]["avg_kernel_us"]))
    print("%-22s %8.2f us  %8.1f M env-steps/s  %s frac %.4f  cpu %s" % (r["name"], k, r.get("env_steps_per_s", r.get("rows_per_s_forward_backward", 0.0)) / 1e6, r["roofline"]["bound"], r["roofline"]["frac"],
          ("%.3f M" % (r["cpu_baseline"]["value"] / 1e6)) if "value" in r.get("cpu_baseline", {}) else r.get("cpu_baseline")))
PY
