#!/bin/bash
# PMC passes for the step kernel (separate runs per counter group; no trace domains besides kernel-trace)
set -o pipefail
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc/$name -- python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-configs ${BENCH_ARGS} > gpurun_out/pmc/$name.json 2> gpurun_out/pmc/$name.err || { tail -5 gpurun_out/pmc/$name.err; return 1; }
  echo "pass $name done" >> gpurun_out/progress.log
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU || exit 1
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
run grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
python - <<'PY'
import csv, glob, collections
for name in ("sq1","sq2","fetch","write","grbm"):
    files = glob.glob(f"gpurun_out/pmc/{name}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if "step_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in acc.items():
        print(f"{name:6s} {k:28s} n={len(v):4d} mean={sum(v)/len(v):.1f}")
PY
