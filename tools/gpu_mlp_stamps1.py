#!/usr/bin/env python3
"""Diagnostic: in-kernel timeline of the single-phase 128-row policy-layer kernels (PPM_STAMP build), K tile 8, per wave group:
fragment-read issue | DMA issue | wait for the reads | wait for the DMAs (lagging group) | barrier | MFMAs | wait for the DMAs (leading
group) | barrier.  Run on the GPU box: python tools/gpu_mlp_stamps1.py [tile 513|514] [m] [k] [n]"""
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
subprocess.run(["hipcc"] + _lib.HIPCC_FLAGS + ["-DPPM_STAMP=1"] + os.environ.get("PPENV_STAMP_DEFS", "").split() + ["-o", lib] + _lib.SOURCES, check=True)
os.environ["PPENV_LIB"] = lib
_lib.LIB_PATH = lib
tile = sys.argv[1] if len(sys.argv) > 1 else "513"
os.environ["PPENV_MLP_TILE"] = tile
import torch  # noqa: E402
from isaacgym_amd.policy import layer_forward  # noqa: E402

m, kin, n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 4096), (3, 1536), (4, 1024)))
dev = torch.device("cuda", 0)
x = torch.randn(m, 2 * kin, device=dev).half()
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
names = ["reads", "DMA issue", "wait reads", "wait DMA (g1)", "barrier", "MFMAs", "wait DMA (g0)", "barrier"]
print("tile %s, layer [%d x %d] x [%d]^T x2: median cycles (s_memtime; each stamp costs ~100) over the first 256 workgroups, K tile 8" % (tile, m, kin, n))
for g, waves in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
    d = np.diff(t[:, waves, 0:9], axis=2)
    med = np.median(d.reshape(-1, 8), axis=0)
    print(g, " | ".join("%s %4.0f" % (nm, v) for nm, v in zip(names, med)))
print("K tile (stamp 0 of tile 9 - stamp 0 of tile 8):", np.median(t[:, :, 9] - t[:, :, 0]), " whole K loop / tiles:", np.median(t[:, :, 31] - t[:, :, 30]) / (kin // 64),
      " epilogue:", np.median(t[:, :, 29] - t[:, :, 31]))
