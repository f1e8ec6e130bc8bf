#!/usr/bin/env python3
"""Race screen for the backward kernels (run on the GPU box): the weight-gradient kernel (transposed LDS reads behind inline assembly, the forward's
alternating wave groups and counted vmcnt) on the policy's shapes and some ragged ones, every split count it can take, and the input-gradient launch
with its ELU' / column-sum store pass — many launches on small-integer operands, each of which must equal the fp32 product bit for bit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd.policy import dw_workspace_bytes, layer_backward_input, layer_backward_weight  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
total_bad = 0
# (m, n, k, splits): dW [n, k] of a layer k -> n at minibatch m
for m, n, k, splits in [(32768, 1536, 2048, 0), (32768, 512, 512, 0), (8192, 1024, 1536, 0), (8192, 512, 1024, 8), (4096, 2048, 320, 4), (2048, 328, 200, 1), (1024, 72, 520, 16)]:
    gen = torch.Generator(device="cuda").manual_seed(m + n)
    dz = torch.randint(-1, 2, (m, 2 * n), generator=gen, device="cuda").to(torch.float16)
    x = torch.randint(-2, 3, (m, 2 * k), generator=gen, device="cuda").to(torch.float16)
    want = torch.stack([dz[:, j * n:(j + 1) * n].float().t() @ x[:, j * k:(j + 1) * k].float() for j in range(2)])
    assert float(want.abs().max()) < 2 ** 24
    ws = torch.empty(max(dw_workspace_bytes(m, n, k, 2, splits), 16), dtype=torch.uint8, device="cuda")
    dw = torch.empty((2, n, k), device="cuda")
    bad = 0
    for rep in range(reps):
        dw.fill_(-7.0)
        layer_backward_weight(dw, dz, x, batch=2, dz_stride=n, x_stride=k, dw_stride=n * k, m=m, n=n, k=k, splits=splits, workspace=ws)
        bad += int((dw != want).sum())
    total_bad += bad
    print("dW  [%5d x %4d]^T [%5d x %4d] x2, splits %s: %d launches, %d wrong elements" % (m, n, m, k, splits or "auto", reps, bad), flush=True)
for tile in (512, 513, 514, 516):
    os.environ["PPENV_MLP_TILE"] = str(tile)
    for m, n, k in [(8192, 1024, 1024), (4096, 512, 1536), (1000, 328, 192)]:      # dx [m, n] = dz [m, k] . wt [n, k]^T
        gen = torch.Generator(device="cuda").manual_seed(tile + m)
        dz = torch.randint(-1, 2, (m, 2 * k), generator=gen, device="cuda").to(torch.float16)
        wt = torch.randint(-1, 2, (2, n, k), generator=gen, device="cuda").to(torch.float16)
        y = torch.tensor([-0.5, -0.25, 0.5, 2.0], device="cuda")[torch.randint(0, 4, (m, 2 * n), generator=gen, device="cuda")].to(torch.float16)
        want = torch.cat([dz[:, j * k:(j + 1) * k].float() @ wt[j].float().t() for j in range(2)], dim=1) * torch.where(y.float() > 0, 1.0, y.float() + 1.0)
        assert float(want.abs().max()) < 2048
        blocks = (m + 63) // 64
        want_cs = torch.cat([want, want.new_zeros(blocks * 64 - m, 2 * n)]).view(blocks, 64, 2 * n).sum(dim=1)
        dx = torch.empty((m, 2 * n), dtype=torch.float16, device="cuda")
        cs = torch.empty((blocks, 2 * n), device="cuda")
        bad = 0
        for rep in range(reps):
            dx.fill_(-7.0)
            cs.fill_(-7.0)
            layer_backward_input(dx, dz, wt, elu_out=y, colsum_partial=cs, batch=2, dz_stride=k, wt_stride=n * k, dx_stride=n, elu_out_stride=n, colsum_stride=n, m=m, n=n, k=k)
            bad += int((dx.float() != want).sum()) + int((cs != want_cs).sum())
        total_bad += bad
        print("dX tile %d  [%5d x %4d] x [%4d]^T x2 (+ ELU', column sums): %d launches, %d wrong elements" % (tile, m, k, n, reps, bad), flush=True)
print("backward race screen: %d wrong elements in all" % total_bad)
sys.exit(1 if total_bad else 0)
