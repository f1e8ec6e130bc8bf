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
for tile in tuple(int(t) for t in os.environ.get("RACE_TILES", "520,521,512,513,514,515,516,517,518").split(",")):     # 520 / 521: the default ring kernels since round 4
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
# the chained launch (ppenv_mlp_chain_forward): the reference's layers 3-6 at 4096 rows, the same buffers launch after launch
from isaacgym_amd.policy import UNITS, _descriptor, chain_forward, chain_status, chain_workspace  # noqa: E402
os.environ.pop("PPENV_MLP_TILE", None)
m, u = 4096, UNITS
gen = torch.Generator(device="cuda").manual_seed(99)
x = torch.randint(-1, 2, (m, 2 * u[1]), generator=gen, device="cuda").to(torch.float16)
ws = {i: torch.randint(-1, 2, (2, u[i], u[i - 1]), generator=gen, device="cuda").to(torch.float16) * (1.0 / 64) for i in range(2, 6)}      # small enough to stay exact through four layers
h = {1: x, **{i: torch.empty(m, 2 * u[i], dtype=torch.float16, device="cuda") for i in range(2, 6)}}
kw = lambda i: dict(out=h[i], x=h[i - 1], w=ws[i], bias=None, elu=False, batch=2, in_stride=u[i - 1], w_stride=u[i] * u[i - 1], bias_stride=0, out_stride=u[i], m=m, n=u[i], k=u[i - 1])
wsp = chain_workspace(m, 2, 4, "cuda")
chain_forward([_descriptor(**kw(i)) for i in range(2, 6)], wsp)
torch.cuda.synchronize()
want = {i: h[i].clone() for i in range(2, 6)}
ref = x.float()
for i in range(2, 6):                       # against fp32 torch once (fp16 rounding of each layer's output included)
    ref = torch.cat([ref[:, j * u[i - 1]:(j + 1) * u[i - 1]] @ ws[i][j].float().t() for j in range(2)], dim=1).half().float()
bad = int((want[5].float() != ref).sum())
for rep in range(reps):
    for i in range(2, 6):
        h[i].fill_(7.0)
    chain_forward([_descriptor(**kw(i)) for i in range(2, 6)], wsp)
    bad += sum(int((h[i] != want[i]).sum()) for i in range(2, 6))
total_bad += bad + chain_status(wsp)
print("chained layers 3-6 [4096 rows]: %d launches, %d wrong elements, status %d" % (reps, bad, chain_status(wsp)), flush=True)
print("race screen: %d wrong elements in all" % total_bad)
sys.exit(1 if total_bad else 0)
