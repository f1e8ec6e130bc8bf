#!/usr/bin/env python3
"""Experiment: the N envs of one GPU as K independent handles of N/K envs, each stepping on its own stream (one HIP graph holds the
K chains of 32 launches, forked once and joined once).  A launch ends with its slowest workgroup; K short chains overlap each
other's tails.  Prints aggregate env-steps/s for K = 1, 2, 4, 8."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from isaacgym_amd import scene  # noqa: E402
from isaacgym_amd.env import PPEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
HORIZON = 32
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for K in (1, 2, 4, 8):
    cnt = N // K
    envs = [PPEnv(scene.build_config("TT", num_envs=cnt, seed=0, env_id_offset=k * cnt), device=dev) for k in range(K)]
    gen = torch.Generator(device=dev).manual_seed(0)
    pools = [[(torch.rand(cnt, 7, device=dev, generator=gen) * 2 - 1).contiguous() for _ in range(8)] for _ in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    for s in range(HORIZON):
        for k in range(K):
            envs[k].step(pools[k][s & 7])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream()
        for k in range(K):
            streams[k].wait_stream(cur)
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                for s in range(HORIZON):
                    envs[k].step(pools[k][s & 7])
        for k in range(K):
            cur.wait_stream(streams[k])
    torch.cuda.synchronize()
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    reps = 64
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * HORIZON)
    print(f"N={N} K={K}: {dt * 1e6:.2f} us per step of all envs, {N / dt / 1e6:.1f} M env-steps/s", flush=True)
    for e in envs:
        e.close()
