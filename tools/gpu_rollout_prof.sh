#!/bin/bash
# rocprofv3 summary of the rollout loop with the native forward: per-kernel time of the policy layers next to the env step
for v in TA:4096 TT:16384; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rollout_${v%%:*} -- python tools/rollout_bench.py --variant ${v%%:*} --num-envs ${v##*:} --policy native --steps 640 > gpurun_out/rollout_${v%%:*}_prof.json 2> gpurun_out/prof_rollout.err || { tail -20 gpurun_out/prof_rollout.err; exit 1; }
  python - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_rollout_${v%%:*}/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("${v%%:*} rollout, kernel time by kernel (rocprofv3 --kernel-trace --stats):")
for r in rows[:12]:
    print("  %5.1f %%  %8.1f us avg  x%-6s %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110]))
PY
done
