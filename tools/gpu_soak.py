#!/usr/bin/env python3
"""Soak: the HIP step kernel against the CPU oracle (test infrastructure) on more seeds than the test suite, every step restarted from
the oracle's state; envs the SensitivityProbe flags are skipped as in the tests.  Prints every tolerance violation and a total.
Run on the GPU box: python tools/gpu_soak.py [steps] [seeds...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import oracle.binding as ob  # noqa: E402
from helpers import RTOL, SCALES, SensitivityProbe  # noqa: E402
from isaacgym_amd import scene  # noqa: E402
from isaacgym_amd.env import PPEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seeds = [int(a) for a in sys.argv[2:]] or [21, 22, 23]
ob.build()
tot = bad = 0
DR = os.environ.get("PPENV_SOAK_DR") == "1"
for variant in (("TT", "T3", "TN") if DR else ("TT", "T3", "TN", "T4")):
    for seed in seeds:
        n = 2048
        A = 2 if variant == "T4" else 1
        cfg = scene.build_config(variant, num_envs=n, seed=seed)
        o = ob.OracleEnv(cfg, threads=16)
        env = PPEnv(scene.build_config(variant, num_envs=n, seed=seed), device="cuda:0")
        probe = SensitivityProbe(ob, cfg)
        rng = np.random.default_rng(seed)
        if DR:   # domain randomisation on (N4): per-env tables, action / observation noise, another gravity — same tables on both sides
            tabs = dict(dof_stiffness_scale=rng.uniform(0.5, 1.5, (7, n)).astype(np.float32), dof_damping_scale=rng.uniform(0.5, 1.5, (7, n)).astype(np.float32),
                        link_mass_scale=rng.uniform(0.5, 1.5, (7, n)).astype(np.float32), restitution_scale=rng.uniform(0.0, 0.7, n).astype(np.float32),
                        friction_scale=rng.uniform(0.7, 1.3, n).astype(np.float32))
            kw = dict(action_noise_sigma=0.02, observation_noise_sigma=0.002)
            for x in (o, probe.o2, env):
                x.set_randomization(**tabs, **kw)
                x.set_gravity(-9.8 - 0.3)
        for t in range(steps):
            a = rng.uniform(-1.2, 1.2, (n * A, 7)).astype(np.float32)
            st = o.get_state()
            env.set_state(st)
            o.step(a)
            env.step(torch.from_numpy(a).cuda())
            keep = ~probe.sensitive(st, a, o)
            got = {k: getattr(env, k).cpu().numpy() for k in ("dof_pos", "dof_vel", "ball", "rew_buf", "reset_buf")}
            checks = [("dof_pos", got["dof_pos"], o.dof_pos, SCALES["dof_pos"]), ("dof_vel", got["dof_vel"], o.dof_vel, SCALES["dof_vel"]),
                      ("ball_pos", got["ball"][0:3], o.ball[0:3], SCALES["ball_pos"]), ("ball_vel", got["ball"][7:10], o.ball[7:10], SCALES["ball_vel"]),
                      ("ball_spin", got["ball"][10:13], o.ball[10:13], SCALES["ball_spin"])]
            for name, g, w, sc in checks:
                m = (np.abs(g - w) > RTOL * sc + RTOL * np.abs(w)).any(axis=0) & keep
                if m.any():
                    bad += int(m.sum())
                    print(variant, seed, t, name, np.nonzero(m)[0][:5], float(np.abs(g - w)[:, m].max()), flush=True)
            rows = np.repeat(keep, A)
            m = (np.abs(got["rew_buf"] - o.rew_buf) > A * probe.rew_atol + RTOL * np.abs(o.rew_buf)) & rows   # the power term sums A * 7 dofs
            m |= (got["reset_buf"] != o.reset_buf) & rows
            if m.any():
                bad += int(m.sum())
                print(variant, seed, t, "rew/reset", np.nonzero(m)[0][:5], flush=True)
            tot += int(keep.sum())
        env.close()
        print(variant, seed, "done", flush=True)
print("domain randomisation on:" if DR else "", "env-steps compared", tot, "violations", bad)
