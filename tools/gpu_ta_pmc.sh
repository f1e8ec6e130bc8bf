#!/bin/bash
# SQ counters and HBM traffic of the 27-DoF step kernel (separate --pmc passes, kernel-trace only); KERNEL / WAVES name the kernel
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
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
KERNEL=${KERNEL:-ta_chain_kernel} WAVES=${WAVES:-384} python - <<'PY'
import csv, glob, collections, os, json
key, waves = os.environ["KERNEL"], float(os.environ["WAVES"])
means = {}
lines = []
for name in ("sq1", "sq2", "fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_ta/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if key in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        means[k] = sum(v) / len(v)
        lines.append(f"{name},{k},{len(v)},{means[k]:.1f},{means[k] / waves:.1f}")
open("gpurun_out/pmc_ta/summary.csv", "w").write("pass,counter,dispatches,mean_per_dispatch,per_wave\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
    t = {"kernel": key, "num_envs": 4096, "FETCH_SIZE_KB": means["FETCH_SIZE"], "WRITE_SIZE_KB": means["WRITE_SIZE"],
         "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of streamed read bytes, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; separate --pmc passes with --kernel-trace only",
         "hbm_bytes_per_launch": int(round((2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024))}
    json.dump(t, open("gpurun_out/pmc_ta/traffic.json", "w"), indent=1)
    print(json.dumps(t))
if "SQ_WAVE_CYCLES" in means:
    wc = means["SQ_WAVE_CYCLES"]
    print("per wave: VALU-busy %.1f %%, waiting on a counter %.1f %%, issue-stalled %.1f %% of its cycles" %
          (100 * means["SQ_ACTIVE_INST_VALU"] / wc, 100 * means["SQ_WAIT_ANY"] / wc, 100 * means["SQ_WAIT_INST_ANY"] / wc))
PY
