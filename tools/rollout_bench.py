#!/usr/bin/env python3
"""Rollout-loop throughput of BASELINE config 5's per-GPU slice: the native 27-DoF env step together with the reference's policy
forward, on one GPU — NOT the bench.py metric.  The policy has the reference's architecture
(cfg/train/HumanoidPingpongTiltG1PPO.yaml:11-30,50-52: separate actor and critic MLPs [2048, 1536, 1024, 1024, 512, 512], ELU,
fixed sigma, mixed_precision, normalize_input), random-initialised.  One rollout step = normalise obs -> actor and critic forward ->
sample and clamp the action -> env.step.  --policy native: the hand-written MFMA forward (isaacgym_amd.policy, include/ppenv_policy.h:
obs normalisation fused into the first layer's tile load, bias + ELU on the accumulators); --policy torch: plain PyTorch / hipBLASLt,
fp16 weights and activations.  Prints one JSON line.

    python tools/rollout_bench.py [--variant TA|TT] [--num-envs 4096] [--steps 320] [--no-graph] [--policy native|torch]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

UNITS = [2048, 1536, 1024, 1024, 512, 512]
HORIZON = 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="TA", choices=["TA", "TT"])
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--policy", default="native", choices=["native", "torch"])
    ap.add_argument("--no-fused-input", dest="fused_input", action="store_false",
                    help="TA: keep the separate normalise-and-pad launch instead of letting the step kernel write the policy's first-layer input itself "
                         "(ppenv_ta_sim_set_policy_input; round 4's pair-wise path: 204.6 against 207.2 us per rollout step, same box — the default since)")
    args = ap.parse_args()

    import torch
    from isaacgym_amd import _lib, scene
    _lib.lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n = args.num_envs
    if args.variant == "TA":
        from isaacgym_amd.tensor_api import TAEnv
        env = TAEnv(n, device=dev, seed=0)
        num_obs, num_act = 313, 27
        step, obs_buf = env.step, env.state.obs_buf
    else:
        from isaacgym_amd.env import PPEnv
        env = PPEnv(scene.build_config("TT", num_envs=n, seed=0), device=dev)
        num_obs, num_act = 80, 7
        step, obs_buf = env.step, env.obs_buf

    def mlp(n_out):
        layers, d = [], num_obs
        for u in UNITS:
            layers += [torch.nn.Linear(d, u), torch.nn.ELU()]
            d = u
        layers.append(torch.nn.Linear(d, n_out))
        return torch.nn.Sequential(*layers).to(dev)

    torch.manual_seed(0)
    actor, critic = mlp(num_act), mlp(1)
    mean = torch.zeros(num_obs, device=dev)
    inv_std = torch.ones(num_obs, device=dev)
    native = None
    attached = False
    if args.policy == "native":
        from isaacgym_amd.policy import NativeMLP, sample_actions
        lin = lambda net: [(m.weight, m.bias) for m in net if isinstance(m, torch.nn.Linear)]
        native = NativeMLP(lin(actor), lin(critic), num_obs, dev, mean=mean, var=torch.ones(num_obs, device=dev) - 1e-5, max_rows=n)
        if args.variant == "TA" and args.fused_input and env.sim.kernel == "chain":
            native.attach_env(env)       # the step kernel writes the normalised fp16 rows itself: no normalise-and-pad launch
            attached = True
    actor, critic = actor.half(), critic.half()   # weights cast once (rl_games' autocast re-casts the fp32 master weights in every call)
    sigma = torch.ones(num_act, device=dev)          # fixed_sigma, const_initializer 0 -> exp(0)
    values = torch.zeros(n, 1, device=dev)

    @torch.no_grad()
    def forward():
        if native is not None:
            return native.forward(obs_buf, prepared=attached)
        x = torch.clamp((obs_buf - mean) * inv_std, -5.0, 5.0).half()     # rl_games RunningMeanStd in eval mode
        return actor(x).float(), critic(x).float()

    action_buf = torch.zeros(n, num_act, device=dev)
    neglogp = torch.zeros(n, device=dev)
    counter = [0]

    @torch.no_grad()
    def rollout_step():
        if native is not None:   # the value is already where the learner reads it (a view of the heads' output); the heads launch also samples, clamps and scores
            counter[0] += 1
            native.forward(obs_buf, prepared=attached, sample=dict(actions=action_buf, sigma=sigma, seed=0, counter=counter[0], neglogp=neglogp))
            step(action_buf)
            return
        mu, v = forward()
        values.copy_(v)
        action = torch.clamp(mu + sigma * torch.randn_like(mu), -1.0, 1.0).contiguous()
        step(action)

    @torch.no_grad()
    def policy_only():
        forward()

    def timed(fn, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    for _ in range(args.warmup):
        rollout_step()
    graph = None
    if not args.no_graph:
        try:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(HORIZON):
                    rollout_step()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001 - report and fall back to eager launches
            print(f"graph capture of the rollout horizon failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            graph = None
    k = max(1, args.steps // HORIZON)
    if graph is not None:
        t_roll = timed(graph.replay, k) / HORIZON
    else:
        t_roll = timed(rollout_step, k * HORIZON)
    t_pol = timed(policy_only, 50)
    acts = torch.rand(n, num_act, device=dev) * 2 - 1
    t_env = timed(lambda: step(acts), 200)
    flops = 2 * 2 * n * sum(a * b for a, b in zip([num_obs] + UNITS, UNITS + [0]) if b)   # two nets; the small heads are left out
    MFMA_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense fp16 / bf16
    print(json.dumps({
        "what": "rollout loop (env step + policy forward) on one GPU; context for BASELINE config 5, not the bench.py metric",
        "variant": args.variant, "num_envs": n, "launch": "HIP graph of 32 rollout steps" if graph is not None else "eager",
        "rollout_env_steps_per_s": n / t_roll, "us_per_rollout_step": t_roll * 1e6,
        "us_policy_forward_eager": t_pol * 1e6, "policy_tflops_eager": flops / t_pol / 1e12,
        "policy_mfma_frac_of_dense_peak": flops / t_pol / 1e12 / MFMA_PEAK_TFLOPS, "policy_gflop_per_step": flops / 1e9,
        "us_env_step_eager": t_env * 1e6,
        "policy": "actor + critic MLP [2048,1536,1024,1024,512,512] ELU, fp16 operands / fp32 accumulation, random init; "
                  + ("hand-written MFMA kernels (normalise-and-pad pass + 7 layer launches + one sampling launch)" if native is not None else "PyTorch / hipBLASLt"),
    }))


if __name__ == "__main__":
    main()
