#!/usr/bin/env python3
"""Step time with domain randomisation ON (all five tables + both noises, cfg/task/HumanoidPingpongTiltG1.yaml:100-169) next to the plain
step, for both table-reading instantiations: the two-wave schedule (round 3, default) and the one-wave kernel (PPENV_STEP_KERNEL=fused).
Run on the GPU box:  python tools/gpu_dr_time.py [N]   (each schedule in its own process: the env reads PPENV_STEP_KERNEL at create)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n):
    import numpy as np
    import torch
    from isaacgym_amd import scene
    from isaacgym_amd.env import PPEnv
    env = PPEnv(scene.build_config("TT", num_envs=n, seed=1), device="cuda:0")
    env.reset_all()
    acts = [torch.rand(n, 7, device="cuda") * 2 - 1 for _ in range(8)]

    def timed(reps=400):
        for t in range(64):
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
    env.set_randomization(dof_stiffness_scale=u(0.5, 1.5, (7, n)), dof_damping_scale=u(0.5, 1.5, (7, n)), link_mass_scale=u(0.5, 1.5, (7, n)),
                          restitution_scale=u(0.0, 0.7, n), friction_scale=u(0.7, 1.3, n))
    tables = timed()
    env.set_randomization(dof_stiffness_scale=u(0.5, 1.5, (7, n)), dof_damping_scale=u(0.5, 1.5, (7, n)), link_mass_scale=u(0.5, 1.5, (7, n)),
                          restitution_scale=u(0.0, 0.7, n), friction_scale=u(0.7, 1.3, n), action_noise_sigma=0.02, observation_noise_sigma=0.002)
    noisy = timed()
    print("TT n=%d schedule=%-5s  eager launches, us per step:  plain %.2f   DR tables %.2f   DR tables + action / observation noise %.2f   status %d" %
          (n, os.environ.get("PPENV_STEP_KERNEL", "split"), plain, tables, noisy, env.status))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
        for sched in ("split", "fused"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=dict(os.environ, PPENV_STEP_KERNEL=sched), check=True)
