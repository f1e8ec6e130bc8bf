#!/usr/bin/env python3
"""Time the native rollout collector (isaacgym_amd/collector.py) on the 27-DoF task: horizons of 32 steps with the reference's network,
eager launches (the sampler's counter is a kernel argument, so the loop is not replayed as a graph), buffers in rl_games' experience
layout.  Run on the GPU box: python tools/collector_bench.py [num_envs] [horizons]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd.collector import RolloutCollector  # noqa: E402
from isaacgym_amd.policy import NativeMLP, UNITS  # noqa: E402
from isaacgym_amd.tensor_api import TAEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
env = TAEnv(n, device=dev, seed=0)
torch.manual_seed(0)


def mlp(n_out):
    d, out = 313, []
    for u in UNITS + [n_out]:
        lin = torch.nn.Linear(d, u)
        out.append((lin.weight, lin.bias))
        d = u
    return out


net = NativeMLP(mlp(27), mlp(1), 313, dev, mean=torch.zeros(313), var=torch.ones(313), max_rows=n)
col = RolloutCollector(env, net, horizon=32, sigma=torch.full((27,), 0.135))
for _ in range(3):
    col.collect().next_horizon()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    col.collect().next_horizon()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"what": "native rollout collector, 27-dof task: 32-step horizons into [horizon, num_envs, ...] buffers (obs, actions, neglogp, mu, values, "
                          "rewards, dones) + bootstrap value + GAE, eager launches", "num_envs": n, "horizon": 32,
                  "ms_per_horizon": dt * 1e3, "us_per_step": dt * 1e6 / 32, "env_steps_per_s": n * 32 / dt,
                  "finite": bool(torch.isfinite(col.advantages).all() and torch.isfinite(col.obs).all())}))
