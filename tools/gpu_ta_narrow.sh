#!/bin/bash
# Round 3: the 27-dof chain-wave kernel with 32 / 16 envs per workgroup (-DTA_ENVS_PER_WG): 2x / 4x the workgroups, idle upper lanes — does a narrower workgroup
# shorten the chain?  Per variant: parity (the chain-kernel test), kernel time at 4096 / 16384 envs, SQ counters per wave, the stamped timeline.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/narrow; rm -rf $out; mkdir -p $out
log=gpurun_out/r03_ta_narrow.txt; : > $log
for epw in 64 32 16; do
  python - $epw <<'PY' || exit 1
import subprocess, sys
from isaacgym_amd import _lib
subprocess.run(["hipcc"] + _lib.HIPCC_FLAGS + ["-DTA_ENVS_PER_WG=" + sys.argv[1], "-o", "gpurun_out/narrow/lib%s.so" % sys.argv[1]] + _lib.SOURCES, check=True)
PY
  export PPENV_LIB=$PWD/$out/lib$epw.so
  echo "== envs per workgroup $epw" | tee -a $log
  timeout -k 10 400 python -m pytest tests/test_ta_physics.py -m gpu -q -k "chain_kernel_step" -p no:cacheprovider 2>&1 | tail -2 | tee -a $log
  for n in 4096 16384; do
    timeout -k 10 300 python bench.py --variant TA --num-envs $n --steps 1024 --warmup 128 --no-cpu-baseline > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
    python -c "
import json; d=json.load(open('$out/b.json')); print('envs/wg $epw  n=$n  kernel %.2f us  %.1f M env-steps/s' % (d['roofline']['avg_kernel_us'], d['value']/1e6))" | tee -a $log
  done
  KERNEL=ta_chain_kernel WAVES=$((6 * 4096 / epw)) bash tools/gpu_ta_pmc.sh > $out/pmc$epw.txt 2>&1 || { tail -5 $out/pmc$epw.txt; exit 1; }
  grep -E "SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_ACTIVE_INST_VALU|SQ_INSTS_VALU|SQ_WAVES|hbm_bytes" $out/pmc$epw.txt | tee -a $log
  PPENV_STAMP_DEFS=-DTA_ENVS_PER_WG=$epw timeout -k 10 300 python tools/gpu_ta_chain_stamps.py 4096 2>/dev/null | tail -4 | tee -a $log
  unset PPENV_LIB
done
