#!/usr/bin/env python3
"""Closed-loop soak: the native policy (random-initialised network of the reference's architecture) drives the env for many steps —
forward, sampling, fused step, no host sync in the loop; every `check` steps the buffers are checked for finiteness and physical
bounds and the device status word for 0.  Run on the GPU box: python tools/gpu_rollout_soak.py [TA|TT] [num_envs] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd import scene  # noqa: E402
from isaacgym_amd.policy import NativeMLP, UNITS, sample_actions  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "TA"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
dev = torch.device("cuda", 0)
if variant == "TA":
    from isaacgym_amd.tensor_api import TAEnv
    env = TAEnv(n, device=dev, seed=0)
    num_obs, num_act, obs_buf = 313, 27, env.state.obs_buf
else:
    from isaacgym_amd.env import PPEnv
    env = PPEnv(scene.build_config(variant, num_envs=n, seed=0), device=dev)
    num_obs, num_act, obs_buf = 80, 7, env.obs_buf
torch.manual_seed(0)


def mlp(n_out):
    d, out = num_obs, []
    for u in UNITS + [n_out]:
        lin = torch.nn.Linear(d, u)
        out.append((lin.weight, lin.bias))
        d = u
    return out


net = NativeMLP(mlp(num_act), mlp(1), num_obs, dev, mean=torch.zeros(num_obs), var=torch.ones(num_obs), max_rows=n)
sigma = torch.full((num_act,), 0.3, device=dev)
actions = torch.zeros(n, num_act, device=dev)
neglogp = torch.zeros(n, device=dev)
check, resets, bad = 500, 0, 0
for t in range(1, steps + 1):
    mu, value = net.forward(obs_buf)
    sample_actions(actions, mu, sigma, 1, t, -1.0, 1.0, neglogp)
    env.step(actions)
    if t % check == 0:
        torch.cuda.synchronize()
        status = env.sim.status if variant == "TA" else env.status
        tensors = {"obs": obs_buf, "rew": env.rew_buf, "mu": mu, "value": value, "neglogp": neglogp}
        ok = status == 0 and all(bool(torch.isfinite(v).all()) for v in tensors.values())
        ok = ok and float(obs_buf.abs().max()) < 1e4 and float(actions.abs().max()) <= 1.0
        bad += 0 if ok else 1
        resets += int(env.reset_buf.sum())
        if not ok or t % (10 * check) == 0:
            print(f"step {t}: {'ok' if ok else 'VIOLATION'}  status {status}  |obs|max {float(obs_buf.abs().max()):.2f}  mean reward {float(env.rew_buf.mean()):.3f}  "
                  f"mean progress {float(env.progress_buf.float().mean()):.1f}", flush=True)
print(f"{variant}: {steps} closed-loop steps x {n} envs = {steps * n / 1e6:.1f} M env-steps, checks failed: {bad}, resets seen at check points: {resets}")
sys.exit(1 if bad else 0)
