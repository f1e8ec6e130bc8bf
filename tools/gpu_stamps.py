#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the step kernel (PP_STAMP build).  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "libppenv_stamp.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-signed-zeros", "-ffinite-math-only",
                "-fPIC", "-shared", "-DPP_STAMP=1", "-o", lib, os.path.join(ROOT, "isaacgym_amd", "csrc", "ppenv.hip"), os.path.join(ROOT, "isaacgym_amd", "csrc", "ppenv_ta.hip"),
                os.path.join(ROOT, "isaacgym_amd", "csrc", "ppenv_ta_sim.hip")], check=True)
os.environ["PPENV_LIB"] = lib
import torch  # noqa: E402
from isaacgym_amd import _lib, scene  # noqa: E402
from isaacgym_amd.env import PPEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variant = sys.argv[2] if len(sys.argv) > 2 else "TT"
env = PPEnv(scene.build_config(variant, num_envs=n, seed=0), device="cuda:0")
gen = torch.Generator(device="cuda").manual_seed(0)
pool = [(torch.rand(n * env.num_agents, 7, device="cuda", generator=gen) * 2 - 1) for _ in range(8)]
for s in range(300):
    env.step(pool[s & 7])
torch.cuda.synchronize()
L = _lib.lib()
nb = (n + 63) // 64
buf = np.zeros(nb * 32, np.uint64)
L.ppenv_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.ppenv_debug_read_stamps(buf.ctypes.data, buf.size) == 0
t = buf.reshape(nb, 32).astype(np.int64)
if os.environ.get("PPENV_STEP_KERNEL", "split") == "fused":
    names = {0: "start", 1: "loads done", 2: "FK0", 3: "arm substep 1", 4: "FK1", 5: "ball substep 1", 6: "arm substep 2", 7: "FK2 (+bodies)",
             8: "ball substep 2", 9: "reward/reset/obs", 10: "stores + obs flush"}
    tot = np.median(t[:, 10] - t[:, 0])
    print(f"N={n}: median wave lifetime between first and last stamp: {tot:.0f} shader cycles")
    for k in range(1, 11):
        d = np.median(t[:, k] - t[:, k - 1])
        print(f"  {names[k]:24s} {d:9.0f} cycles  {100 * d / tot:5.1f} %")
else:
    t0 = np.minimum(t[:, 0], t[:, 16])
    arm = {1: "arm: loads + targets", 2: "arm: substep 1 (vel sweep + ABA) + publish", 4: "arm: substep 2 + publish",
           5: "arm: FK of final state + publish paddle", 6: "arm: body obs", 7: "arm: final barrier wait"}
    ball = {17: "ball: loads + next serve", 18: "ball: (no wait)", 19: "ball: FK + substep 1", 20: "ball: wait for boundary 1",
            21: "ball: FK + substep 2", 22: "ball: wait final + reward/reset/obs tail", 23: "ball: final barrier wait", 24: "ball: flush + stores"}
    tot = np.median(t[:, 24] - t0)
    print(f"N={n}: median workgroup span (first stamp -> last stamp): {tot:.0f} shader cycles")
    for k, name in arm.items():
        prev = t[:, k - 1] if k != 4 else t[:, 2]
        print(f"  {name:34s} {np.median(t[:, k] - prev):8.0f} cycles   (ends at {np.median(t[:, k] - t0):6.0f})")
    for k, name in ball.items():
        prev = t[:, k - 1]
        print(f"  {name:34s} {np.median(t[:, k] - prev):8.0f} cycles   (ends at {np.median(t[:, k] - t0):6.0f})")
