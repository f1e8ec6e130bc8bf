#!/bin/bash
# rocprofv3 kernel trace of the 27-DoF task step (TAEnv.step) at BASELINE config 5's per-GPU size
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/ta_loop.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from isaacgym_amd.tensor_api import TAEnv
n = 4096
env = TAEnv(n, device="cuda:0")
gen = torch.Generator(device="cuda").manual_seed(0)
pool = [torch.rand(n, 27, device="cuda", generator=gen) * 2 - 1 for _ in range(8)]
for s in range(400): env.step(pool[s & 7])
torch.cuda.synchronize()
PY
rm -rf gpurun_out/prof_ta
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ta -- python /tmp/ta_loop.py > gpurun_out/prof_ta.log 2>&1 || { tail -5 gpurun_out/prof_ta.log; exit 1; }
for f in $(find gpurun_out/prof_ta -name "*kernel_stats.csv"); do head -8 $f; done
