#!/bin/bash
# same-box A/B of the 27-dof chain kernel: HEAD's library (build_variants/ta_base) against the working tree's
set -o pipefail
[ -f build_variants/ta_base/libppenv.so ] || { echo "build_variants/ta_base/libppenv.so is missing: build the baseline first, in the container — git stash; python -c \"from isaacgym_amd import _lib; _lib.build(out='build_variants/ta_base/libppenv.so', force=True)\"; git stash pop"; exit 2; }
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for lib in base new; do
  if [ $lib = base ]; then export PPENV_LIB=$PWD/build_variants/ta_base/libppenv.so; else unset PPENV_LIB; fi
  timeout -k 10 300 python bench.py --variant TA --num-envs 4096 --steps 1024 --warmup 128 --no-cpu-baseline --no-configs > gpurun_out/bench_TA_ab.json 2>/dev/null || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/bench_TA_ab.json')); print('$lib rep $rep  TA 4096: %.2f us' % d['roofline']['avg_kernel_us'])" | tee -a gpurun_out/ta_ab.txt
  for att in 0 1; do PIPE_ATTACH=$att timeout -k 10 200 python tools/gpu_rollout_pipeline.py 4096 1 2>&1 | grep "^N=" | sed "s/^/$lib attach=$att  /" | cut -c1-90 | tee -a gpurun_out/ta_ab.txt; done
done; done
