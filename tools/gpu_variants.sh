#!/bin/bash
# bench lines + rocprofv3 kernel stats of the variants that are not the headline workload (parity-test cases of BASELINE.json):
# the 4-actor step at config 4's per-GPU size, the 27-dof step at config 5's
set -o pipefail
mkdir -p gpurun_out/variants
export TMPDIR=/tmp
for spec in "T4 8192" "TA 4096"; do
  set -- $spec
  timeout -k 10 400 python bench.py --steps 1000 --warmup 100 --variant $1 --num-envs $2 > gpurun_out/variants/bench_$1.json 2> gpurun_out/variants/bench_$1.err || { tail -20 gpurun_out/variants/bench_$1.err; exit 1; }
  rm -rf gpurun_out/variants/prof_$1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/variants/prof_$1 -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --variant $1 --num-envs $2 > gpurun_out/variants/bench_prof_$1.json 2> gpurun_out/variants/prof_$1.err || { tail -20 gpurun_out/variants/prof_$1.err; exit 1; }
  python - "$1" <<'PY'
import json, sys, glob, csv
v = sys.argv[1]
d = json.load(open(f"gpurun_out/variants/bench_{v}.json"))
print("%s value %.1f M env-steps/s  kernel %.2f us  frac %.4f  cpu %.0f env-steps/s on %d cores" % (v, d["value"]/1e6, d["roofline"]["avg_kernel_us"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
f = sorted(glob.glob(f"gpurun_out/variants/prof_{v}/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:3]:
    print("   ", r["Name"][:70], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
