#!/bin/bash
# Round 3: ball waves per 64 envs (step_kernel_split's BW = PPENV_BALL_WAVES): kernel time per variant / size, then the stamped timeline.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/ballwaves.txt
for bw in 1 2 4; do
  for spec in "TT 16384" "TT 4096" "TT 65536" "T3 16384" "TN 16384"; do
    set -- $spec
    PPENV_BALL_WAVES=$bw timeout -k 10 300 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline --no-configs --variant $1 --num-envs $2 > gpurun_out/bench_bw.json 2> gpurun_out/bench_bw.err || { tail -20 gpurun_out/bench_bw.err; exit 1; }
    python - $bw "$@" <<'PY' | tee -a gpurun_out/ballwaves.txt
import json, sys
d=json.load(open("gpurun_out/bench_bw.json"))
print("ball_waves=%s %s n=%-6s value %8.1f M env-steps/s  kernel %7.2f us" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"]/1e6, d["roofline"]["avg_kernel_us"]))
PY
  done
done
for bw in 1 2 4; do
  echo "== stamps, ball_waves=$bw" | tee -a gpurun_out/ballwaves.txt
  PPENV_BALL_WAVES=$bw timeout -k 10 300 python tools/gpu_stamps.py 16384 TT 2>/dev/null | tee -a gpurun_out/ballwaves.txt
done
