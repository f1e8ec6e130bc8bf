#!/bin/bash
# SQ counters of the 27-DoF rigid-body kernel (separate --pmc passes, kernel-trace only)
set -o pipefail
mkdir -p gpurun_out/pmc_ta
export TMPDIR=/tmp
cat > /tmp/ta_loop.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from isaacgym_amd.tensor_api import TAEnv
n = 4096
env = TAEnv(n, device="cuda:0")
gen = torch.Generator(device="cuda").manual_seed(0)
pool = [torch.rand(n, 27, device="cuda", generator=gen) * 2 - 1 for _ in range(8)]
for s in range(100): env.step(pool[s & 7])
torch.cuda.synchronize()
PY
run() { name=$1; shift
  rm -rf gpurun_out/pmc_ta/$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_ta/$name -- python /tmp/ta_loop.py > gpurun_out/pmc_ta/$name.log 2>&1 || { tail -5 gpurun_out/pmc_ta/$name.log; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU || exit 1
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_FLAT || exit 1
python - <<'PY'
import csv, glob, collections
for name in ("sq1","sq2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_ta/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "ta_sim_quad_kernel<true" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(f"{name} {k:24s} n={len(v)} mean per dispatch {sum(v)/len(v):14.1f}   per wave {sum(v)/len(v)/256:10.1f}")
PY
