"""Experiment: does replaying a captured HIP graph of 32 steps shorten the per-step time (launch-to-launch gap)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaacgym_amd import scene
from isaacgym_amd.env import PPEnv
n = 16384
env = PPEnv(scene.build_config("TT", num_envs=n, seed=0), device="cuda:0")
gen = torch.Generator(device="cuda").manual_seed(0)
pool = [(torch.rand(n, 7, device="cuda", generator=gen) * 2 - 1).contiguous() for _ in range(8)]
for s in range(200): env.step(pool[s & 7])
torch.cuda.synchronize()
def timeit(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
def eager(reps):
    for s in range(reps): env.step(pool[s & 7])
print("eager  %.2f us/step" % (timeit(eager, 1920) / 1920))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for s in range(32): env.step(pool[s & 7])
torch.cuda.synchronize()
def replay(reps):
    for _ in range(reps // 32): g.replay()
replay(64)
print("graph  %.2f us/step" % (timeit(replay, 1920) / 1920))
