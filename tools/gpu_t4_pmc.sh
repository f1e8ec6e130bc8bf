#!/bin/bash
# SQ counters and HBM traffic of the 4-actor step kernel at BASELINE config 4's per-GPU size (8192 envs): separate --pmc passes, kernel-trace only
# (no *_T4_pmc_traffic.json existed, roofline.traffic of the T4 row was null).  -> gpurun_out/pmc_t4/summary.csv, traffic.json
set -o pipefail
mkdir -p gpurun_out/pmc_t4
export TMPDIR=/tmp
N=${T4_ENVS:-8192}
run() { name=$1; shift
  rm -rf gpurun_out/pmc_t4/$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_t4/$name -- python bench.py --variant T4 --num-envs $N --steps 200 --warmup 50 --no-cpu-baseline --no-configs > gpurun_out/pmc_t4/$name.json 2> gpurun_out/pmc_t4/$name.err || { tail -5 gpurun_out/pmc_t4/$name.err; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU || exit 1
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
T4_ENVS=$N python - <<'PY'
import csv, glob, collections, os, json
n = int(os.environ["T4_ENVS"])
means, lines, kernel = {}, [], ""
for name in ("sq1", "sq2", "fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_t4/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "step_kernel_split" in row["Kernel_Name"]:
                kernel = row["Kernel_Name"]
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        means[k] = sum(v) / len(v)
        lines.append(f"{name},{k},{len(v)},{means[k]:.1f}")
open("gpurun_out/pmc_t4/summary.csv", "w").write("pass,counter,dispatches,mean_per_dispatch\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
short = kernel.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip() if kernel else "step_kernel_split<pp::ModelG1, 2, 1, 1, false>"
t = {"kernel": short, "num_envs": n, "FETCH_SIZE_KB": means["FETCH_SIZE"], "WRITE_SIZE_KB": means["WRITE_SIZE"],
     "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of streamed read bytes, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; separate --pmc passes with --kernel-trace only",
     "hbm_bytes_per_launch": int(round((2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024))}
json.dump(t, open("gpurun_out/pmc_t4/traffic.json", "w"), indent=1)
print(json.dumps(t))
wc = means["SQ_WAVE_CYCLES"]
print("per wave: VALU-busy %.1f %%, waiting %.1f %% of its cycles" % (100 * means["SQ_ACTIVE_INST_VALU"] / wc, 100 * means["SQ_WAIT_ANY"] / wc))
PY
