#!/bin/bash
# A/B of differently-built libraries (build_variants/libppenv_<name>.so, built on the CPU side): TT / T4 / TA bench lines per build
set -o pipefail
mkdir -p gpurun_out
for lib in build_variants/libppenv_*.so; do
  for spec in ${SPECS:-"TT 16384" "TT 65536" "T4 8192" "TA 4096"}; do
    set -- $spec
    PPENV_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline --variant $1 --num-envs $2 > gpurun_out/bench_lv.json 2> gpurun_out/bench_lv.err || { tail -20 gpurun_out/bench_lv.err; exit 1; }
    python - "$lib" "$1" "$2" <<'PY'
import json, sys
d=json.load(open("gpurun_out/bench_lv.json"))
print("%-40s %s n=%-6s value %8.1f M env-steps/s  kernel %7.2f us" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"]/1e6, d["roofline"]["avg_kernel_us"]))
PY
  done
done
