#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
run() { name=$1; shift
  rm -rf gpurun_out/pmc/$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc/$name -- python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-configs ${BENCH_ARGS} > gpurun_out/pmc/$name.json 2> gpurun_out/pmc/$name.err || { tail -5 gpurun_out/pmc/$name.err; return 1; }
}
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY || exit 1
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS || exit 1
python - <<'PY'
import csv, glob, collections
for name in ("ic","dc"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "step_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    w = sum(acc["SQ_WAVES"])/len(acc["SQ_WAVES"])
    for k,v in sorted(acc.items()):
        m=sum(v)/len(v); print(f"{name} {k:28s} total {m:12.1f} per-wave {m/w:10.1f}")
PY
