#!/usr/bin/env python3
"""Experiment / measurement: the config-5 rollout (4096 envs of the 27-dof task + the native policy forward) as K
independent env GROUPS of 4096 / K envs, each group's own policy -> env -> policy chain on its own stream inside ONE captured graph
(forked once, joined once per 32-step horizon).  The groups share the weights (NativeMLP.sibling).  While one group's env step
(64-CU-class, latency-bound) or narrow layer runs, the other groups' MFMA layers fill the idle CUs.  Prints us per rollout step of ALL
envs for K = 1, 2, 4 (K = 1 is bench.py's rollout row)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from isaacgym_amd.policy import NativeMLP  # noqa: E402
from isaacgym_amd.tensor_api import TAEnv  # noqa: E402

UNITS = [2048, 1536, 1024, 1024, 512, 512]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
# configurations "K" or "K:cus" — cus: the CU share each group's layer launches are sized for (ppenv_mlp_layer_forward_share; 0 = the chip)
KS = [tuple(int(x) for x in (k.split(":") + ["0"])[:2]) for k in (sys.argv[2] if len(sys.argv) > 2 else "1,2,2:128,4,4:64").split(",")]
ATTACH = os.environ.get("PIPE_ATTACH", "1") == "1"     # the env's step kernel writes the normalised fp16 rows (no normalise-and-pad launch)
HORIZON = 32
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
out = {}
for K, CUS in KS:
    cnt = N // K
    envs = [TAEnv(cnt, device=dev, seed=0, env_id_offset=k * cnt) for k in range(K)]
    net0 = NativeMLP(actor, critic, num_obs, dev, mean=torch.zeros(num_obs, device=dev), var=torch.ones(num_obs, device=dev) - 1e-5, max_rows=cnt, cus=CUS)
    nets = [net0] + [net0.sibling(cnt) for _ in range(K - 1)]
    sigma = torch.ones(num_act, device=dev)
    acts = [torch.zeros(cnt, num_act, device=dev) for _ in range(K)]
    nlps = [torch.zeros(cnt, device=dev) for _ in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    if ATTACH:
        for k in range(K):
            nets[k].attach_env(envs[k])

    def chain(k, s):
        nets[k].forward(envs[k].obs_buf, prepared=ATTACH, sample=dict(actions=acts[k], sigma=sigma, seed=k, counter=s + 1, neglogp=nlps[k]))
        envs[k].step(acts[k])

    with torch.no_grad():
        for s in range(4):
            for k in range(K):
                chain(k, s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            if K == 1:
                for s in range(HORIZON):
                    chain(0, s)
            else:
                for k in range(K):
                    streams[k].wait_stream(cur)
                for k in range(K):
                    with torch.cuda.stream(streams[k]):
                        for s in range(HORIZON):
                            chain(k, s)
                for k in range(K):
                    cur.wait_stream(streams[k])
        torch.cuda.synchronize()
        for _ in range(6):
            g.replay()
        torch.cuda.synchronize()
        res = []
        for _ in range(5):
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / (reps * HORIZON) * 1e6)
    us = sorted(res)[len(res) // 2]
    ok = all(bool(torch.isfinite(e.obs_buf).all()) and e.sim.status == 0 for e in envs)
    out[f"K={K},cus={CUS}"] = {"envs_per_group": cnt, "cus_per_group": CUS or 256, "env_writes_policy_input": ATTACH, "us_per_rollout_step_all_envs": round(us, 2), "env_steps_per_s": round(N / us * 1e6), "finite_and_healthy": ok}
    print(f"N={N} K={K} cus={CUS or 256}: {us:.1f} us per rollout step of all {N} envs = {N / us:.2f} M env-steps/s  (regions: {[round(r, 1) for r in res]})", flush=True)
    for e in envs:
        e.close()
    del nets, net0, envs, g
    torch.cuda.empty_cache()
print(json.dumps({"rollout_groups": out, "num_envs": N}))
