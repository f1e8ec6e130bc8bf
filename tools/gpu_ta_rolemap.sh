#!/bin/bash
# Round 3: which roles of the 27-dof chain-wave kernel share a SIMD (-DTA_ROLE_MAP): kernel time at 4096 / 16384 envs and the stamped timeline per placement.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/rolemap; rm -rf $out; mkdir -p $out
log=gpurun_out/r03_ta_rolemap.txt; : > $log
for rm in 0 1 2; do
  python - $rm <<'PY' || exit 1
import subprocess, sys
from isaacgym_amd import _lib
subprocess.run(["hipcc"] + _lib.HIPCC_FLAGS + ["-DTA_ROLE_MAP=" + sys.argv[1], "-o", "gpurun_out/rolemap/lib%s.so" % sys.argv[1]] + _lib.SOURCES, check=True)
PY
  export PPENV_LIB=$PWD/$out/lib$rm.so
  echo "== role map $rm" | tee -a $log
  timeout -k 10 400 python -m pytest tests/test_ta_physics.py -m gpu -q -k "chain_kernel_step" -p no:cacheprovider 2>&1 | tail -1 | tee -a $log
  for n in 4096 16384; do
    timeout -k 10 300 python bench.py --variant TA --num-envs $n --steps 1024 --warmup 128 --no-cpu-baseline > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
    python -c "
import json; d=json.load(open('$out/b.json')); print('role map $rm  n=$n  kernel %.2f us  %.1f M env-steps/s' % (d['roofline']['avg_kernel_us'], d['value']/1e6))" | tee -a $log
  done
  TA_ROLE_MAP=$rm PPENV_STAMP_DEFS=-DTA_ROLE_MAP=$rm timeout -k 10 300 python tools/gpu_ta_chain_stamps.py 4096 2>/dev/null | grep -v amdgpu | tee -a $log
  unset PPENV_LIB
done
