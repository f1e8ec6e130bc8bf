#!/usr/bin/env python3
"""Per-layer timing of the native policy forward (and torch.nn.functional.linear on the same shapes).  Run on the GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd.policy import UNITS, layer_forward, prepare_input  # noqa: E402

dev = torch.device("cuda", 0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 313


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


dims = [k0] + UNITS
obs = torch.randn(m, k0, device=dev)
mean, istd = torch.zeros(k0, device=dev), torch.ones(k0, device=dev)
prev = None
tot_n = tot_t = 0.0
for i, (kin, n) in enumerate(zip(dims[:-1], dims[1:])):
    kp = (kin + 63) // 64 * 64                                  # rows padded to a K tile, as NativeMLP.load does
    w = torch.zeros(2, n, kp, device=dev, dtype=torch.float16)
    w[:, :, :kin] = (torch.randn(2, n, kin, device=dev) / kin ** 0.5).half()
    b = torch.zeros(2, n, device=dev).half()
    out = torch.empty(m, 2 * n, device=dev, dtype=torch.float16)
    if i == 0:
        x16p = torch.empty(m, kp, device=dev, dtype=torch.float16)
        fused = lambda: layer_forward(out, obs, w.view(2 * n, kp), b.view(-1), elu=True, mean=mean, inv_std=istd, k=kin)
        prep = lambda: prepare_input(x16p, obs, mean, istd)
        gemm = lambda: layer_forward(out, x16p, w.view(2 * n, kp), b.view(-1), elu=True)
        print("layer 1 fused (fp32 obs staged by the layer kernel) %.1f us; split: normalise + pad %.1f us, layer %.1f us" % (timeit(fused), timeit(prep), timeit(gemm)))
        fn = lambda: (prep(), gemm())
        x16 = obs.half()
        wt = w[:, :, :kin].reshape(2 * n, kin).contiguous()
        ft = lambda: torch.nn.functional.elu(torch.nn.functional.linear(x16, wt, b.view(-1)))
    else:
        x = prev
        fn = lambda: layer_forward(out, x, w, b, elu=True, batch=2, in_stride=kin, w_stride=n * kin, bias_stride=n, out_stride=n, m=m, n=n, k=kin)
        ft = lambda: [torch.nn.functional.elu(torch.nn.functional.linear(x[:, j * kin:(j + 1) * kin], w[j][:, :kin], b[j])) for j in range(2)]
    tn, tt = timeit(fn), timeit(ft)
    fl = 2 * 2 * m * kin * n
    tot_n += tn
    tot_t += tt
    print("layer %d: [%d x %d] x [%d]^T x2  native %7.1f us %6.0f TF   torch(linear+elu) %7.1f us %6.0f TF" % (i + 1, m, kin, n, tn, fl / tn / 1e6, tt, fl / tt / 1e6))
    prev = out
print("PPENV_MLP_TILE=%s" % os.environ.get("PPENV_MLP_TILE", "auto"), end=" ")
print("sum native %.1f us, torch %.1f us" % (tot_n, tot_t))
