#!/usr/bin/env python3
"""Race screen for the LDS-DMA policy kernels (run on the GPU box): every tile configuration, several shapes, many launches, small-integer
operands — each launch must equal the fp32 matmul bit for bit.  Prints one line per (configuration, shape)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd.policy import layer_forward  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shapes = [(16384, 1536, 2048), (4096, 1536, 2048), (4096, 1024, 1536), (4096, 512, 512), (16384, 512, 1024), (1000, 600, 192), (300, 1536, 64)]
total_bad = 0
for tile in (512, 513, 514, 516, 517):
    os.environ["PPENV_MLP_TILE"] = str(tile)
    for m, n, k in shapes:
        gen = torch.Generator(device="cuda").manual_seed(tile + m)
        a = torch.randint(-1, 2, (m, 2 * k), generator=gen, device="cuda").to(torch.float16)
        w = torch.randint(-2, 3, (2, n, k), generator=gen, device="cuda").to(torch.float16)
        bias = torch.randint(-2, 3, (2, n), generator=gen, device="cuda").to(torch.float16)
        want = torch.cat([a[:, j * k:(j + 1) * k].float() @ w[j].float().t() + bias[j].float() for j in range(2)], dim=1)
        out = torch.empty((m, 2 * n), dtype=torch.float32, device="cuda")
        bad = 0
        for rep in range(reps):
            out.fill_(-7.0)
            layer_forward(out, a, w, bias, elu=False, batch=2, in_stride=k, w_stride=n * k, bias_stride=n, out_stride=n, m=m, n=n, k=k)
            bad += int((out != want).sum())
        total_bad += bad
        print("tile %d  [%5d x %4d] x [%4d]^T x2: %d launches, %d wrong elements" % (tile, m, k, n, reps, bad), flush=True)
print("race screen: %d wrong elements in all" % total_bad)
sys.exit(1 if total_bad else 0)
