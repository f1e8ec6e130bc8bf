#!/bin/bash
# GPU iteration for the 27-DoF variant: its tests, then a timing of TAEnv.step at BASELINE config 5's per-GPU size
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ta_physics.py tests/test_ta_golden.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_ta.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_ta.log; [ $rc -ne 0 ] && { tail -60 gpurun_out/pytest_ta.log; exit $rc; }
timeout -k 10 300 python - <<'PY'
import time, torch
from isaacgym_amd.tensor_api import TAEnv
import os
for mapping, n in (("chain", 4096), ("quad", 4096), ("chain", 16384), ("chain", 65536), ("quad", 16384)):
    os.environ["PPENV_TA_KERNEL"] = mapping
    env = TAEnv(n, device="cuda:0")
    gen = torch.Generator(device="cuda").manual_seed(0)
    pool = [torch.rand(n, 27, device="cuda", generator=gen) * 2 - 1 for _ in range(8)]
    for s in range(50): env.step(pool[s & 7])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K = 300
    for s in range(K): env.step(pool[s & 7])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / K
    print("TA %s n=%d  step %.1f us  %.1f M env-steps/s" % (mapping, n, us, n / us))
    env.close()
PY
