#!/usr/bin/env python3
"""Experiment: the config-5 rollout with the ACTOR in front of the env step and the CRITIC beside both — two streams inside one captured graph.  Only
mu -> actions are needed by the env step; the value is needed by the learner at the end of the horizon.  Stream A (optionally high priority): actor layers,
heads + draw, env step.  Stream B: critic layers + value head, started when step s's policy input exists (after env step s - 1) and joined before the input
is overwritten (env step s).  Prints us per rollout step for: the batched forward (bench.py's row), the split on one stream, the split on two streams with
and without a priority for stream A; and checks mu / value / actions against the batched forward bit for bit."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from isaacgym_amd.policy import NativeMLP  # noqa: E402
from isaacgym_amd.tensor_api import TAEnv  # noqa: E402

UNITS = [2048, 1536, 1024, 1024, 512, 512]
N, HORIZON = 4096, 32
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
num_obs, num_act = 313, 27


def mlp(n_out):
    layers, d = [], num_obs
    for u in UNITS:
        layers += [torch.nn.Linear(d, u), torch.nn.ELU()]
        d = u
    layers.append(torch.nn.Linear(d, n_out))
    return [(m.weight, m.bias) for m in torch.nn.Sequential(*layers) if isinstance(m, torch.nn.Linear)]


torch.manual_seed(0)
actor, critic = mlp(num_act), mlp(1)
sigma = torch.ones(num_act, device=dev)


def build(mode):
    env = TAEnv(N, device=dev, seed=0)
    net = NativeMLP(actor, critic, num_obs, dev, mean=torch.zeros(num_obs, device=dev), var=torch.ones(num_obs, device=dev) - 1e-5, max_rows=N)
    net.attach_env(env)
    acts, nlp = torch.zeros(N, num_act, device=dev), torch.zeros(N, device=dev)
    prio = -1 if mode == "two_streams_priority" else 0
    sa, sb = torch.cuda.Stream(device=dev, priority=prio), torch.cuda.Stream(device=dev)
    smp = lambda s: dict(actions=acts, sigma=sigma, seed=0, counter=s + 1, neglogp=nlp)

    def step(s, cur):
        if mode == "batched":
            net.forward(env.obs_buf, prepared=True, sample=smp(s))
            env.step(acts)
        elif mode == "split_one_stream":
            net.forward_net(0, env.obs_buf, prepared=True, sample=smp(s))
            env.step(acts)
            net.forward_net(1, env.obs_buf, prepared=True)          # (reads the NEXT observation's input: timing only)
        else:
            with torch.cuda.stream(sb):
                net.forward_net(1, env.obs_buf, prepared=True)
            with torch.cuda.stream(sa):
                net.forward_net(0, env.obs_buf, prepared=True, sample=smp(s))
                sa.wait_stream(sb)                                    # the env step rewrites the policy input the critic's first layer reads (conservative: the whole critic chain)
                env.step(acts)
            sb.wait_stream(sa)

    def run_eager(k):
        cur = torch.cuda.current_stream()
        if mode.startswith("two"):
            sa.wait_stream(cur); sb.wait_stream(cur)
        for s in range(k):
            step(s, cur)
        if mode.startswith("two"):
            cur.wait_stream(sa); cur.wait_stream(sb)
    return env, net, acts, run_eager


out = {}
ref = None
for mode in ("batched", "split_one_stream", "two_streams", "two_streams_priority"):
    env, net, acts, run_eager = build(mode)
    with torch.no_grad():
        run_eager(4)
        torch.cuda.synchronize()
        if mode == "batched":
            ref = (net.head_out.clone(), acts.clone())
        elif mode != "split_one_stream":
            same = torch.equal(net.head_out, ref[0]) and torch.equal(acts, ref[1])
            print(f"{mode}: mu | value and actions after 4 steps equal the batched forward's bit for bit: {same}", flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            run_eager(HORIZON)
        torch.cuda.synchronize()
        for _ in range(6):
            g.replay()
        torch.cuda.synchronize()
        res = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / (10 * HORIZON) * 1e6)
    us = sorted(res)[2]
    ok = bool(torch.isfinite(env.obs_buf).all()) and env.sim.status == 0
    print(f"{mode:22s}: {us:6.1f} us per rollout step  ({N / us:.2f} M env-steps/s)  healthy {ok}", flush=True)
    env.close()
    del net, env, g
    torch.cuda.empty_cache()
