#!/usr/bin/env python3
"""The reference's own surface, timed: `isaacgym_amd.make(task=...)` -> `task.step(actions)` called from Python like rl_games' env wrapper does
(train.py:122-167), eager launches, no graph — what a user of the unchanged training script gets per call.  python tools/vec_task_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import isaacgym_amd  # noqa: E402

for name, n in (("HumanoidPingpongTiltG1", 16384), ("HumanoidPingpongTiltG1", 4096), ("Humanoid12PingpongTiltG1", 8192), ("HumanoidPingpongTiltNESSparse27DOFG1", 4096)):
    task = isaacgym_amd.make(seed=1, task=name, num_envs=n, sim_device="cuda:0", rl_device="cuda:0")
    acts = [(torch.rand(task.num_envs * getattr(task, "num_agents", 1), task.num_actions, device="cuda:0") * 2 - 1) for _ in range(8)]
    task.reset()
    for s in range(200):
        task.step(acts[s & 7])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 2000
    for s in range(steps):
        obs, rew, done, info = task.step(acts[s & 7])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:40s} num_envs {n:6d}: {dt / steps * 1e6:7.1f} us per VecTask.step call = {n * steps / dt / 1e6:8.1f} M env-steps/s (eager, host-driven)", flush=True)
    del task
