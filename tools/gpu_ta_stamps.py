#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the fused 27-dof step kernel (TA_STAMP build).  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "libppenv_tastamp.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
src = os.path.join(ROOT, "isaacgym_amd", "csrc")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine", "-fno-signed-zeros", "-ffinite-math-only",
                "-fPIC", "-shared", "-DTA_STAMP=1", "-o", lib, os.path.join(src, "ppenv.hip"), os.path.join(src, "ppenv_ta.hip"), os.path.join(src, "ppenv_ta_sim.hip")], check=True)
os.environ["PPENV_LIB"] = lib
import torch  # noqa: E402
from isaacgym_amd import _lib  # noqa: E402
from isaacgym_amd.tensor_api import TAEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = TAEnv(n, device="cuda:0", seed=0)
gen = torch.Generator(device="cuda").manual_seed(0)
pool = [(torch.rand(n, 27, device="cuda", generator=gen) * 2 - 1) for _ in range(8)]
for s in range(200):
    env.step(pool[s & 7])
torch.cuda.synchronize()
L = _lib.lib()
nb = (n + 15) // 16
buf = np.zeros(nb * 32, np.uint64)
L.ppenv_ta_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.ppenv_ta_debug_read_stamps(buf.ctypes.data, buf.size) == 0
t = buf.reshape(nb, 32).astype(np.int64)
names = [(1, 0, "table copies + barrier + base load"), (2, 1, "substep 1: pass 1 (kinematics, link dynamics, contacts)"), (3, 2, "substep 1: pass 2 (articulated inertias)"),
         (4, 3, "substep 1: hub merge + base solve"), (5, 4, "substep 1: pass 3 (accelerations, integration)"), (6, 5, "substep 1: ball (one lane per quad)"),
         (7, 6, "substep 2: pass 1"), (8, 7, "substep 2: pass 2"), (9, 8, "substep 2: hub merge + base solve"), (10, 9, "substep 2: pass 3"), (11, 10, "substep 2: ball"),
         (16, 11, "output: FK of the final state + link rows (quaternions) into the LDS tile"), (12, 16, "output: root / table / ball rows, dof tiles"), (13, 12, "rigid-body tile -> HBM"), (14, 13, "task arithmetic (reward, reset, 313 obs)"),
         (15, 14, "obs / root / dof tiles -> HBM")]
tot = t[:, 15] - t[:, 0]
print(f"N={n}: workgroup span after the table copies, percentiles 0/50/90/100: " + " ".join(f"{np.percentile(tot, q):.0f}" for q in (0, 50, 90, 100)))
for a, b, nm in names:
    d = t[:, a] - t[:, b]
    print(f"  {nm:70s} {np.median(d):8.0f} cycles  {100 * np.median(d) / np.median(tot):5.1f} %   (max {d.max():.0f})")
