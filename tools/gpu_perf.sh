#!/bin/bash
# perf iteration: bench line + SQ counters (no parity tests)
set -o pipefail
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/bench_quick.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_quick.json"))
print("value %.1f M env-steps/s  kernel %.2f us  frac %.4f" % (d["value"]/1e6, d["roofline"]["avg_kernel_us"], d["roofline"]["frac"]))
PY
rm -rf gpurun_out/pmc/sq1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/pmc/sq1 -- python bench.py --steps 200 --warmup 50 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/pmc/sq1.json 2> gpurun_out/pmc/sq1.err || { tail -5 gpurun_out/pmc/sq1.err; exit 1; }
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc/sq1/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "step_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
w = sum(acc["SQ_WAVES"])/len(acc["SQ_WAVES"])
for k,v in sorted(acc.items()):
    m=sum(v)/len(v); print(f"{k:24s} per-wave {m/w:10.1f}" + ("  (x4 = %.0f cycles)"%(4*m/w) if "CYCLES" in k or "WAIT" in k or "ACTIVE" in k else ""))
PY
