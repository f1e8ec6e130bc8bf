#!/usr/bin/env python3
"""Diagnostic: in-kernel timeline of the chain-wave 27-dof step (TA_STAMP build), per wave role.  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd import _lib  # noqa: E402
lib = os.path.join(ROOT, "gpurun_out", "libppenv_chainstamp.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.run(["hipcc"] + _lib.HIPCC_FLAGS + ["-DTA_STAMP=1"] + os.environ.get("PPENV_STAMP_DEFS", "").split() + ["-o", lib] + _lib.SOURCES, check=True)
os.environ["PPENV_LIB"] = lib
_lib.LIB_PATH = lib
import torch  # noqa: E402
from isaacgym_amd.tensor_api import TAEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
os.environ["PPENV_TA_KERNEL"] = "chain"
env = TAEnv(n, device="cuda:0", seed=0)
gen = torch.Generator(device="cuda").manual_seed(0)
pool = [(torch.rand(n, 27, device="cuda", generator=gen) * 2 - 1) for _ in range(8)]
for s in range(200):
    env.step(pool[s & 7])
torch.cuda.synchronize()
L = _lib.lib()
nb = min((n + 63) // 64, 1024)
buf = np.zeros(1024 * 6 * 32, np.uint64)
L.ppenv_ta_chain_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.ppenv_ta_chain_debug_read_stamps(buf.ctypes.data, buf.size) == 0
t = buf.reshape(1024, 6, 32)[:nb].astype(np.int64)
t0 = t[:, :, 0].min(axis=1, keepdims=True)[:, :, None]
rel = t - t0
names = {"0": ["waist", "right leg", "left arm", "right arm", "left leg", "ball"], "1": ["waist", "right leg", "left leg", "left arm", "right arm", "ball"],
         "2": ["waist", "right leg", "left leg", "right arm", "left arm", "ball"]}[os.environ.get("TA_ROLE_MAP", "1")]   # wave index order (ppenv_ta_chain.hip W_*, per TA_ROLE_MAP)
pts = {0: "start", 1: "inputs staged", 2: "s1 begin", 3: "s1 pass1 done", 7: "s1 (waist) pelvis dyn", 4: "s1 pass2 done / arms in", 8: "s1 (waist) waist pass2", 9: "s1 (waist) legs in",
       5: "s1 accel in/out", 6: "s1 end", 10: "s2 begin", 11: "s2 pass1 done", 15: "s2 (waist) pelvis dyn", 12: "s2 pass2 done / arms in", 16: "s2 (waist) waist pass2",
       17: "s2 (waist) legs in", 13: "s2 accel in/out", 14: "s2 end", 20: "B1 arrive", 21: "B1 leave", 22: "B2 arrive", 23: "B2 leave", 24: "B3 arrive", 25: "B3 leave", 27: "obs rest stored", 28: "dof_states stored", 29: "dof_force stored", 30: "root_states stored", 26: "end"}
order = [0, 1, 2, 3, 7, 4, 8, 9, 5, 6, 10, 11, 15, 12, 16, 17, 13, 14, 20, 21, 22, 23, 24, 25, 27, 28, 29, 30, 26]
print(f"N={n}: {nb} workgroups; median cycles since the workgroup's first stamp (s_memtime, 100 MHz-independent shader clock)")
print("%-28s" % "point" + "".join("%11s" % x for x in names))
for k in order:
    row = np.median(rel[:, :, k], axis=0)
    used = (t[:, :, k] != 0).any(axis=0)
    print("%-28s" % pts[k] + "".join(("%11.0f" % row[w]) if used[w] else "%11s" % "-" for w in range(6)))
span = rel[:, :, 26].max(axis=1)
print("workgroup span percentiles 0/50/90/100:", " ".join("%.0f" % np.percentile(span, q) for q in (0, 50, 90, 100)))
