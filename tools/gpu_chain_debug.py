#!/usr/bin/env python3
"""Debug: the chained-layer launch on growing problem sizes, each under its own wall-clock print, with the workspace dumped afterwards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from isaacgym_amd.policy import _descriptor, chain_forward, chain_status, chain_workspace, layer_forward

def run(m, dims):
    gen = torch.Generator(device="cuda").manual_seed(m)
    n_layers = len(dims) - 1
    w = [(torch.randn(2, dims[i + 1], dims[i], device="cuda", generator=gen) / dims[i] ** 0.5).half() for i in range(n_layers)]
    b = [(torch.randn(2, dims[i + 1], device="cuda", generator=gen) * 0.1).half() for i in range(n_layers)]
    x = torch.randn(m, 2 * dims[0], device="cuda", generator=gen).half()
    mk = lambda fill: [x] + [torch.full((m, 2 * dims[i + 1]), fill, dtype=torch.float16, device="cuda") for i in range(n_layers)]
    ha, hb = mk(7.0), mk(-7.0)
    kw = lambda i, h: dict(out=h[i + 1], x=h[i], w=w[i], bias=b[i], elu=True, batch=2, in_stride=dims[i], w_stride=dims[i + 1] * dims[i], bias_stride=dims[i + 1],
                           out_stride=dims[i + 1], m=m, n=dims[i + 1], k=dims[i])
    for i in range(n_layers):
        os.environ["PPENV_MLP_TILE"] = "521" if ((dims[i + 1] + 255) // 256) * ((m + 127) // 128) * 2 >= 192 else "520"
        layer_forward(**kw(i, ha))
    os.environ.pop("PPENV_MLP_TILE")
    torch.cuda.synchronize()
    print(f"m={m}: per-layer launches done", flush=True)
    ws = chain_workspace(m, 2, n_layers, "cuda")
    for rep in range(3):
        t0 = time.perf_counter()
        print("  launching", flush=True)
        chain_forward([_descriptor(**kw(i, hb)) for i in range(n_layers)], ws)
        print("  launched", flush=True)
        torch.cuda.synchronize()
        print("  synchronised", flush=True)
        dt = time.perf_counter() - t0
        bad = [int((ha[i + 1] != hb[i + 1]).sum()) for i in range(n_layers)]
        print(f"m={m} dims={dims} rep {rep}: {dt * 1e3:.2f} ms  status {chain_status(ws)}  mismatching elements per layer {bad}  ws[:3]={ws[:3].tolist()} counters max {int(ws[3:].max()) if ws.numel() > 3 else 0}", flush=True)

cases = {"tiny": ((128, [256, 256, 256]),), "all": ((128, [256, 256, 256]), (256, [512, 512, 512]), (1024, [1024, 512, 512]), (4096, [1024, 512, 512]), (4096, [1536, 1024, 1024, 512, 512]))}
for m, dims in cases[sys.argv[1] if len(sys.argv) > 1 else "all"]:
    run(m, dims)
