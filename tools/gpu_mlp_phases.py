#!/usr/bin/env python3
"""Diagnostic (PPM_STAMP build): where one layer launch of the policy forward spends its time at the rollout's M = 4096 — per layer shape of
the reference's network, with the tile the library picks: launch duration (HIP events, graph of 20 launches), and from in-kernel stamps of the
first 256 workgroups (median over workgroups, shader cycles): prologue (kernel entry -> first K tile ready), K loop, epilogue; from the
100 MHz real-time clock: first workgroup entry -> last workgroup exit (the in-kernel span) and the spread of entries / exits.
    python tools/gpu_mlp_phases.py [M]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd import _lib  # noqa: E402
lib = os.path.join(ROOT, "build_variants", "ppstamp", "libppenv.so")       # built in the container beforehand (ships with the snapshot), else here
_lib.build(out=lib, extra_flags=["-DPPM_STAMP=1"])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
os.environ["PPENV_LIB"] = lib
_lib.LIB_PATH = lib
import torch  # noqa: E402
from isaacgym_amd.policy import layer_forward  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
L = _lib.lib()
L.ppenv_mlp_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
dims = [320, 2048, 1536, 1024, 1024, 512, 512]
seen = 0
print(f"M = {m}; cycles = shader clock (s_memtime), us = 100 MHz real-time clock (s_memrealtime) / HIP events")
for i in range(6):
    kin, n = dims[i], dims[i + 1]
    batch = 1 if i == 0 else 2
    if i == 0:
        x = torch.randn(m, kin, device=dev).half()
        w = (torch.randn(2 * n, kin, device=dev) / kin ** 0.5).half()
        b = torch.zeros(2 * n, device=dev).half()
        out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
        run = lambda: layer_forward(out, x, w, b, elu=True)
    else:
        x = torch.randn(m, 2 * kin, device=dev).half()
        w = (torch.randn(2, n, kin, device=dev) / kin ** 0.5).half()
        b = torch.zeros(2, n, device=dev).half()
        out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
        run = lambda: layer_forward(out, x, w, b, elu=True, batch=2, in_stride=kin, w_stride=n * kin, bias_stride=n, out_stride=n, m=m, n=n, k=kin)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    buf = np.zeros(256 * 8 * 32, np.uint64)
    assert L.ppenv_mlp_debug_read_stamps(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(256, 8, 32).astype(np.int64)
    live = t[:, 0, 27] > seen                     # workgroups of THIS grid (blockIdx.x < 256, blockIdx.y == 0): the buffer keeps older launches' stamps
    seen = int(t[:, :, 27:29].max())
    t = t[live]
    pro, loop, epi = t[:, :, 25] - t[:, :, 26], t[:, :, 31] - t[:, :, 25], t[:, :, 29] - t[:, :, 31]
    rt0, rt1 = t[:, :, 27].min(axis=1), t[:, :, 28].max(axis=1)      # per workgroup: entry, exit (10 ns ticks)
    span = (rt1.max() - rt0.min()) / 100.0
    clk = np.median((t[:, 0, 29] - t[:, 0, 26]) / np.maximum(t[:, 0, 28] - t[:, 0, 27], 1)) * 100 / 1e3
    flops = 2 * m * kin * n * 2 * (1 if i else 1)
    print(f"layer {i + 1}: [{m} x {kin}] x [{n}]^T x2   launch {us:6.2f} us ({flops / us / 1e6:5.0f} TF)   in-kernel span {span:6.2f} us   "
          f"entries spread {(rt0.max() - rt0.min()) / 100.0:5.2f} us, exits spread {(rt1.max() - rt1.min()) / 100.0:5.2f} us   "
          f"per workgroup (median, cycles): prologue {np.median(pro):6.0f}  K loop {np.median(loop):7.0f} ({np.median(loop) / (kin // 64):5.0f} per K tile)  "
          f"epilogue {np.median(epi):6.0f}   clock {clk:4.2f} GHz   stamped workgroups {int(live.sum())}")
    if os.environ.get("PHASES_TILE") and kin // 64 > 9:      # the ring kernel's stamps inside K tile 8: per group, cycles between consecutive stamps
        names = ["reads issued", "DMA issued", "lgkmcnt(0)", "vmcnt (lagging group)", "barrier 1", "MFMAs issued", "vmcnt (leading group)", "barrier 2"]
        for g, waves in (("leading group (waves 0-3)", slice(0, 4)), ("lagging group (waves 4-7)", slice(4, 8))):
            d = np.diff(t[:, waves, 0:9], axis=2).reshape(-1, 8)
            if waves.start == 4:      # where the lagging group is when the leading group starts its MFMAs of the same tile (0 = it starts its reads of that tile)
                print("      lagging group's read start of tile 8 minus: leading group's read start of tile 8 %.0f, leading group's MFMA start of tile 8 %.0f, leading group's read start of tile 9 %.0f"
                      % (np.median(t[:, 4, 0] - t[:, 0, 0]), np.median(t[:, 4, 0] - t[:, 0, 5]), np.median(t[:, 4, 0] - t[:, 0, 9])))
            print("      K tile 8, " + g + ": " + "  ".join(f"{nm} {np.median(d[:, i]):.0f}" for i, nm in enumerate(names)) + f"   | whole tile {np.median(t[:, waves, 9] - t[:, waves, 0]):.0f}")
