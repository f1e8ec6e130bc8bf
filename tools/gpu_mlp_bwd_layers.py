#!/usr/bin/env python3
"""Per-layer timing of the native policy BACKWARD at the learner's minibatch (default M = 32768, cfg/train/HumanoidPingpongTiltG1PPO.yaml:75)
next to PyTorch / hipBLASLt on the same fp16 operands:
    dX  = (dZ . W) * ELU'(y)   native: one launch (forward tile kernels on the transposed weight image, ELU' and the bias-gradient
                               column sums in the store pass)          torch: matmul + aten.elu_backward + a column sum (what autograd launches)
    dW  = dZ^T . X             native: transposed LDS reads, split over M + fixed-order reduce      torch: matmul of the transposes
Run on the GPU box:  python tools/gpu_mlp_bwd_layers.py [M] [num_obs] [--splits S]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd.policy import UNITS, dw_workspace_bytes, layer_backward_input, layer_backward_weight, reduce_rows  # noqa: E402

dev = torch.device("cuda", 0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m = int(args[0]) if len(args) > 0 else 32768
k0 = int(args[1]) if len(args) > 1 else 313
splits = int(sys.argv[sys.argv.index("--splits") + 1]) if "--splits" in sys.argv else 0


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


dims = [(k0 + 63) // 64 * 64] + UNITS
blocks = (m + 63) // 64
tot = dict(dx_n=0.0, dx_t=0.0, dw_n=0.0, dw_t=0.0)
for i in range(len(UNITS) - 1, -1, -1):
    k, n = dims[i], dims[i + 1]
    # activations of an ELU layer: about half negative
    dz = (torch.randn(m, 2 * n, device=dev) * 0.05).half()
    x = torch.nn.functional.elu(torch.randn(m, 2 * k if i else k, device=dev)).half()
    w = (torch.randn(2, n, k, device=dev) / k ** 0.5).half()
    wt = w.transpose(1, 2).contiguous()
    dw = torch.empty(2, n, k, device=dev)
    fl = 2 * 2 * m * n * k
    if i == 0:
        ws = torch.empty(max(dw_workspace_bytes(m, 2 * n, k, 1, splits), 16), dtype=torch.uint8, device=dev)
        f_dw = lambda: layer_backward_weight(dw.view(2 * n, k), dz, x, splits=splits, workspace=ws)
        t_dw = lambda: torch.matmul(dz.t(), x)
        dwn, dwt = timeit(f_dw), timeit(t_dw)
        print("layer 1: dW [%d x %d]^T [%d x %d]      native %7.1f us %5.0f TF   torch %7.1f us %5.0f TF   (no dX: the observations need no gradient)" %
              (m, 2 * n, m, k, dwn, fl / dwn / 1e6, dwt, fl / dwt / 1e6))
        tot["dw_n"] += dwn
        tot["dw_t"] += dwt
        continue
    ws = torch.empty(max(dw_workspace_bytes(m, n, k, 2, splits), 16), dtype=torch.uint8, device=dev)
    dx = torch.empty(m, 2 * k, device=dev, dtype=torch.float16)
    cs = torch.empty(blocks, 2 * k, device=dev)
    db = torch.empty(2 * k, device=dev)
    f_dw = lambda: layer_backward_weight(dw, dz, x, batch=2, dz_stride=n, x_stride=k, dw_stride=n * k, m=m, n=n, k=k, splits=splits, workspace=ws)
    t_dw = lambda: [torch.matmul(dz[:, j * n:(j + 1) * n].t(), x[:, j * k:(j + 1) * k]) for j in range(2)]

    def f_dx():
        layer_backward_input(dx, dz, wt, elu_out=x, colsum_partial=cs, batch=2, dz_stride=n, wt_stride=k * n, dx_stride=k, elu_out_stride=k, colsum_stride=k, m=m, n=k, k=n)
        reduce_rows(db, cs, rows=blocks, n=2 * k)

    def t_dx():     # what autograd launches per network: the matmul, elu_backward on the saved output, the bias gradient's column sum
        out = []
        for j in range(2):
            g = torch.matmul(dz[:, j * n:(j + 1) * n], w[j])
            g = torch.ops.aten.elu_backward(g, 1.0, 1.0, 1.0, True, x[:, j * k:(j + 1) * k])
            out.append((g, g.sum(dim=0)))
        return out
    dwn, dwt, dxn, dxt = timeit(f_dw), timeit(t_dw), timeit(f_dx), timeit(t_dx)
    print("layer %d: %4d -> %4d x2   dX native %7.1f us %5.0f TF  torch %7.1f us %5.0f TF   |   dW native %7.1f us %5.0f TF  torch %7.1f us %5.0f TF" %
          (i + 1, k, n, dxn, fl / dxn / 1e6, dxt, fl / dxt / 1e6, dwn, fl / dwn / 1e6, dwt, fl / dwt / 1e6))
    tot["dx_n"] += dxn
    tot["dx_t"] += dxt
    tot["dw_n"] += dwn
    tot["dw_t"] += dwt
print("M = %d, splits = %s: sum dX native %.0f us / torch %.0f us;  sum dW native %.0f us / torch %.0f us;  backward native %.0f us / torch %.0f us" %
      (m, splits or "auto", tot["dx_n"], tot["dx_t"], tot["dw_n"], tot["dw_t"], tot["dx_n"] + tot["dw_n"], tot["dx_t"] + tot["dw_t"]))
