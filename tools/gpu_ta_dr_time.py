#!/usr/bin/env python3
"""27-dof step time with domain randomisation on (tables; tables + noise) next to the plain chain-wave kernel.  python tools/gpu_ta_dr_time.py [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd.tensor_api import TAEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = TAEnv(n, device="cuda:0", seed=1)
acts = [torch.rand(n, 27, device="cuda") * 2 - 1 for _ in range(8)]


def timed(reps=300):
    for t in range(40):
        env.step(acts[t % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(reps):
        env.step(acts[t % 8])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


plain = timed()
rng = np.random.default_rng(0)
u = lambda lo, hi, shape: rng.uniform(lo, hi, shape).astype(np.float32)
tabs = dict(dof_stiffness_scale=u(0.6, 1.4, (27, n)), dof_damping_scale=u(0.6, 1.4, (27, n)), link_mass_scale=u(0.7, 1.3, (28, n)), restitution_scale=u(0, 0.7, n), friction_scale=u(0.7, 1.3, n))
env.set_randomization(**tabs)
tables = timed()
env.set_randomization(**tabs, action_noise_sigma=0.02, observation_noise_sigma=0.002)
noisy = timed()
print("TA n=%d chain-wave kernel, eager launches, us per step:  plain %.2f   DR tables %.2f   DR tables + action / observation noise %.2f   status %d" % (n, plain, tables, noisy, env.sim.status))
