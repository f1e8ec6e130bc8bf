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
SRC = os.environ.get("PPENV_STAMP_SRC", os.path.join(ROOT, "isaacgym_amd", "csrc"))   # another source tree to stamp (A/B of two builds)
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine", "-fno-signed-zeros", "-ffinite-math-only",
                "-fPIC", "-shared", "-DPP_STAMP=1", *os.environ.get("PPENV_STAMP_DEFS", "").split(), "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(SRC, "ppenv.hip"), os.path.join(SRC, "ppenv_ta.hip"),
                os.path.join(SRC, "ppenv_ta_sim.hip"), os.path.join(SRC, "ppenv_ta_chain.hip"), os.path.join(SRC, "ppenv_policy.hip")], check=True)
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
    if os.environ.get("PPENV_STEP_KERNEL") == "quad":
        arm = {1: "arm: loads + targets", 2: "arm: substep 1 (vel sweep + ABA) + publish", 4: "arm: substep 2 + publish", 5: "arm: (serve draw moved)",
               6: "arm: FK of final state + bodies to LDS", 7: "arm: X, bodies 0-3, Y", 8: "arm: flush share"}
    ball = {17: "ball: loads + next serve", 18: "ball: (no wait)", 19: "ball: FK + substep 1", 20: "ball: wait for boundary 1",
            21: "ball: FK + substep 2", 22: "ball: wait final + reward/reset/obs tail", 23: "ball: final barrier wait", 24: "ball: flush + stores"}
    tot = np.median(t[:, 24] - t0)
    print(f"N={n}: median workgroup span (first stamp -> last stamp): {tot:.0f} shader cycles")
    span = t[:, 24] - t0
    print("  span percentiles 0/10/50/90/99/100: " + " ".join(f"{np.percentile(span, q):.0f}" for q in (0, 10, 50, 90, 99, 100)))
    print(f"  first workgroup start -> last workgroup end: {t[:, 24].max() - t0.min():.0f} cycles; start spread {t0.max() - t0.min():.0f}")
    for a, b, nm in ((2, 1, "arm substep 1"), (4, 2, "arm substep 2"), (19, 18, "ball FK + substep 1"), (21, 20, "ball FK + substep 2"), (22, 21, "ball tail")):
        d = t[:, a] - t[:, b]
        print(f"  {nm:22s} percentiles 0/50/90/100: " + " ".join(f"{np.percentile(d, q):.0f}" for q in (0, 50, 90, 100)))
    # inside the LAST ball substep (slots 25..31: FK done, contact tables built, micro-steps 0..3 start, loop end)
    order = np.argsort(t[:, 21] - t[:, 20])
    groups = (("median 10 %", order[int(0.45 * nb):int(0.55 * nb)]), ("slowest 2 %", order[-max(1, nb // 50):]))
    inner = ((25, 20, "wait + FK sweep"), (26, 25, "per-substep shape tables"), (28, 27, "micro-step 0"), (29, 28, "micro-step 1"), (30, 29, "micro-step 2"),
             (31, 30, "micro-step 3"), (21, 31, "quaternion + publish"))
    for gname, idx in groups:
        print(f"  last ball substep, {gname}: " + ", ".join(f"{nm} {np.median(t[idx, a] - t[idx, b]):.0f}" for a, b, nm in inner))
    for k, name in arm.items():
        prev = t[:, k - 1] if k != 4 else t[:, 2]
        print(f"  {name:34s} {np.median(t[:, k] - prev):8.0f} cycles   (ends at {np.median(t[:, k] - t0):6.0f})")
    for k, name in ball.items():
        prev = t[:, k - 1]
        print(f"  {name:34s} {np.median(t[:, k] - prev):8.0f} cycles   (ends at {np.median(t[:, k] - t0):6.0f})")
