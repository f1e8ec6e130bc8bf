#!/bin/bash
# Round 3: how the 27-dof kernel's 110 KB per workgroup leave the CU (-DTA_FLUSH_MODE=0 plain stores / 1 non-temporal)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fm
for m in 1 0; do
  python - $m <<'PY' || exit 1
import subprocess, sys
from isaacgym_amd import _lib
subprocess.run(["hipcc"] + _lib.HIPCC_FLAGS + ["-DTA_FLUSH_MODE=" + sys.argv[1], "-o", "gpurun_out/fm/lib%s.so" % sys.argv[1]] + _lib.SOURCES, check=True)
PY
  for n in 4096 16384; do
    PPENV_LIB=$PWD/gpurun_out/fm/lib$m.so timeout -k 10 300 python bench.py --variant TA --num-envs $n --steps 1024 --warmup 128 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flush mode $m  n=$n  kernel %.2f us' % d['roofline']['avg_kernel_us'])"
  done
done
