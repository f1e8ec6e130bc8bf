#!/bin/bash
# Round 3: who stores the dof state in the two-wave 7-dof kernel (-DPP_BALL_STORES_DOFS=0: the arm wave, after waiting for the ball wave's reset decision)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ds
for m in 1 0; do
  python - $m <<'PY' || exit 1
import subprocess, sys
from isaacgym_amd import _lib
subprocess.run(["hipcc"] + _lib.HIPCC_FLAGS + ["-DPP_BALL_STORES_DOFS=" + sys.argv[1], "-o", "gpurun_out/ds/lib%s.so" % sys.argv[1]] + _lib.SOURCES, check=True)
PY
  for n in 16384 4096 65536; do
    PPENV_LIB=$PWD/gpurun_out/ds/lib$m.so timeout -k 10 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-configs --num-envs $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ball stores dofs = $m  N=$n  kernel %.2f us  %.1f M env-steps/s' % (d['roofline']['avg_kernel_us'], d['value']/1e6))"
  done
done
PPENV_LIB=$PWD/gpurun_out/ds/lib0.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
