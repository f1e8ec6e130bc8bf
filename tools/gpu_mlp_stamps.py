#!/usr/bin/env python3
"""Diagnostic: in-kernel timeline of the 256 x 256 policy-layer kernel (PPM_STAMP build): cycles per barrier interval of K tiles 8 and 9,
per wave group.  Run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd import _lib  # noqa: E402
lib = os.path.join(ROOT, "gpurun_out", "libppenv_ppstamp.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
exp = os.environ.get("PP_EXP", "0")      # timing experiments, see ppenv_policy.hip
subprocess.run(["hipcc"] + _lib.HIPCC_FLAGS + ["-DPPM_STAMP=1", "-DPP_EXP=" + exp, "-o", lib] + _lib.SOURCES, check=True)
os.environ["PPENV_LIB"] = lib
_lib.LIB_PATH = lib
os.environ["PPENV_MLP_TILE"] = "512"
import torch  # noqa: E402
from isaacgym_amd.policy import layer_forward  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kin, n = 2048, 1536
dev = torch.device("cuda", 0)
x = (torch.randn(m, 2 * kin, device=dev)).half()
w = (torch.randn(2, n, kin, device=dev) / kin ** 0.5).half()
b = torch.zeros(2, n, device=dev).half()
out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
for _ in range(5):
    layer_forward(out, x, w, b, elu=True, batch=2, in_stride=kin, w_stride=n * kin, bias_stride=n, out_stride=n, m=m, n=n, k=kin)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(256 * 8 * 32, np.uint64)
L.ppenv_mlp_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.ppenv_mlp_debug_read_stamps(buf.ctypes.data, buf.size) == 0
t = buf.reshape(256, 8, 32).astype(np.int64)
print("PP_EXP=%s" % exp, end=" ")
print("layer [%d x %d] x [%d]^T: median cycles (s_memtime) over 256 workgroups" % (m, kin, n))
for g, waves in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
    d = np.diff(t[:, waves, 0:5], axis=2)              # tile 8: read A | MFMA A | read B | MFMA B (barrier to barrier)
    med = np.median(d.reshape(-1, 4), axis=0)
    print(g, "tile 8:", " ".join("%5.0f" % v for v in med))
print("K tile (stamp 0 of tile 9 - stamp 0 of tile 8):", np.median(t[:, :, 9] - t[:, :, 0]))
print("whole K loop / 32 tiles:", np.median(t[:, :, 31] - t[:, :, 30]) / 32, " epilogue:", np.median(t[:, :, 29] - t[:, :, 31]))
