#!/usr/bin/env python3
"""Diagnostic: what bounds a K tile of the ring kernel (mlp_layer_pp1_kernel) on the narrow layers at M = 4096 — the same launch timed in
builds with one ingredient of the K loop removed (-DPP_EXP: 1 no fragment reads, 2 no DMA, 4 no barriers, 8 no MFMAs, 16 no epilogue; the
results of those builds are wrong, only their time is looked at).   python tools/gpu_mlp_exp1.py build | run"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXPS = [int(x) for x in os.environ.get("EXPS", "0,1,2,4,8,16,3,11,15,31").split(",")]
libs = {e: os.path.join(ROOT, "build_variants", f"ppexp_{e}", "libppenv.so") for e in EXPS}
if sys.argv[1] == "build":
    from isaacgym_amd import _lib
    for e, lib in libs.items():
        _lib.build(out=lib, extra_flags=[f"-DPP_EXP={e}"])
    sys.exit(0)
if sys.argv[1] == "run":
    for e in EXPS:
        subprocess.run([sys.executable, __file__, "child", str(e)], check=True)
    sys.exit(0)
e = int(sys.argv[2])
os.environ["PPENV_LIB"] = libs[e]
from isaacgym_amd import _lib  # noqa: E402
_lib.LIB_PATH = libs[e]
import torch  # noqa: E402
from isaacgym_amd.policy import layer_forward  # noqa: E402

dev = torch.device("cuda", 0)
m = 4096
res = []
for tile, kin, n in ((514, 1024, 512), (514, 512, 512), (513, 1024, 1024), (513, 1536, 1024), (520, 1024, 512), (521, 1024, 1024), (518, 1024, 512)):
    os.environ["PPENV_MLP_TILE"] = str(tile)
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
    res.append(f"{kin}->{n} tile {tile}: {e0.elapsed_time(e1) * 1e3 / 200:6.2f} us")
names = {1: "no fragment reads", 2: "no DMA", 4: "no barriers", 8: "no MFMAs", 16: "no epilogue", 32: "no s_setprio around the MFMAs"}
what = " + ".join(v for k, v in names.items() if e & k) or "the real kernel"
print(f"PP_EXP={e:2d} ({what}):  " + "   ".join(res), flush=True)
