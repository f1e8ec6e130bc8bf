#!/usr/bin/env python3
"""BASELINE config 5's per-GPU slice end to end, as context (a TOOL, not the bench.py metric and not a trainer shipped by this repository): one PPO
epoch of rl_games' a2c_continuous as the reference configures it (cfg/train/HumanoidPingpongTiltG1PPO.yaml:50-85: horizon 32, mini_epochs 5, e_clip 0.2,
critic_coef 4, clip_value, bounds_loss_coef 1e-4, grad_norm 10, normalize_advantage, mixed_precision, normalize_input) on the 27-dof task at 4096 envs:

    rollout     RolloutCollector: 32 x (native policy forward + action draw + fused env step), bootstrap value, GAE          [native, this repository]
    learning    mini_epochs x (131072 / minibatch) minibatch steps: network forward + backward   --learner native: NativeActorCritic (MFMA kernels)
                                                                                                   --learner torch : nn.Sequential under autocast(fp16)
                PPO losses (PyTorch elementwise on [M, 27] tensors), grad-norm clip, fused Adam   [PyTorch either way]

rl_games itself is not importable here (absent from the reference and the image); the loss terms are restated from its published a2c_continuous /
common_losses (actor: clipped surrogate; critic: clipped value loss; bound loss on mu beyond +-1.1) — timing context, parity unpinned.
Run on the GPU box:  python tools/ppo_epoch_bench.py [--learner native|torch] [--num-envs 4096] [--minibatch 32768] [--epochs 3]"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaacgym_amd.collector import RolloutCollector  # noqa: E402
from isaacgym_amd.policy import NativeActorCritic, UNITS  # noqa: E402
from isaacgym_amd.tensor_api import TAEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--learner", default="native", choices=["native", "torch"])
ap.add_argument("--num-envs", type=int, default=4096)
ap.add_argument("--minibatch", type=int, default=32768)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--force-dist", action="store_true", help="create the nccl group and all-reduce the gradients even with ONE rank (the RCCL path on a one-GPU box)")
args = ap.parse_args()
# data-parallel learners, one rank per GPU (the reference's multi_gpu mode, train.py:117-120): launched by torch.distributed.run, each rank owns num_envs envs
# and its own learner; the gradients are averaged over the ranks before every optimizer step (native learner: one all-reduce per layer beside the backward)
import torch.distributed as dist  # noqa: E402
from isaacgym_amd import distributed as D  # noqa: E402
rank, local_rank, world = D.rank_info()
dev = torch.device("cuda", local_rank)
torch.cuda.set_device(dev)
use_dist = world > 1 or args.force_dist
if use_dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
n, H, A, NOBS = args.num_envs, 32, 27, 313
MINI_EPOCHS, E_CLIP, CRITIC_COEF, BOUNDS_COEF, GRAD_NORM, LR = 5, 0.2, 4.0, 1e-4, 10.0, 2e-5     # yaml:60-85
torch.manual_seed(0)


def mlp(n_out):
    d, out = NOBS, []
    for u in UNITS + [n_out]:
        lin = torch.nn.Linear(d, u)
        out.append((lin.weight.detach(), lin.bias.detach()))
        d = u
    return out


actor, critic = mlp(A), mlp(1)
env = TAEnv(n, device=dev, seed=0, env_id_offset=rank * n)
native = NativeActorCritic(actor, critic, NOBS, dev, normalize_input=True)          # serves the rollout in both modes
logstd = torch.nn.Parameter(torch.zeros(A, device=dev))                             # fixed_sigma: a learnable, observation-independent log-std (yaml:21-27)
col = RolloutCollector(env, native.learner.net, horizon=H, sigma=torch.exp(logstd.detach()))

if args.learner == "torch":
    def seq(layers):
        mods = []
        for i, (w, b) in enumerate(layers):
            lin = torch.nn.Linear(w.shape[1], w.shape[0])
            lin.weight.data.copy_(w); lin.bias.data.copy_(b)
            mods.append(lin)
            if i + 1 < len(layers):
                mods.append(torch.nn.ELU())
        return torch.nn.Sequential(*mods).to(dev)
    t_actor, t_critic = seq(actor), seq(critic)
    params = list(t_actor.parameters()) + list(t_critic.parameters()) + [logstd]

    def net_forward(obs):          # rl_games' model under autocast: normalise (statistics frozen here), two MLPs in fp16, fp32 heads out
        x = torch.clamp((obs - native.learner.rms.mean) * native.learner.rms.inv_std, -5.0, 5.0)
        with torch.autocast("cuda", dtype=torch.float16):
            return t_actor(x).float(), t_critic(x).float()
else:
    params = list(native.parameters()) + [logstd]
    params = [p for p in params if p.requires_grad]
    net_forward = native
    if use_dist:
        native.grad_sync = D.GradientBuckets(force=args.force_dist)
opt = torch.optim.Adam(params, lr=LR, eps=1e-8, fused=True)
scale = 1024.0          # a constant loss scale (GradScaler's job in rl_games): fp16 gradients inside the network either way


def learn():
    """The minibatch loop of one epoch on the collector's buffers -> number of minibatch steps."""
    rows = H * n
    obs = col.obs[:H].reshape(rows, NOBS)
    act, old_v = col.actions.reshape(rows, A), col.values[:H].reshape(rows, 1)
    # the collector keeps the CLAMPED actions (what the env consumed) and the draw's own log-probability; rl_games keeps the unclamped draw.  For a ratio that
    # starts at 1 the old log-probability is re-evaluated on the stored actions under the rollout's mu / sigma (one elementwise pass per epoch)
    with torch.no_grad():
        old_nlp = 0.5 * (((act - col.mu[:H].reshape(rows, A)) / col.sigma) ** 2).sum(dim=1) + 0.5 * math.log(2.0 * math.pi) * A + torch.log(col.sigma).sum()
    ret = col.returns.reshape(rows, 1)
    adv = col.advantages.reshape(rows)
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)                                                   # normalize_advantage
    steps = 0
    native.train()
    for _ in range(MINI_EPOCHS):
        for lo in range(0, rows, args.minibatch):
            sl = slice(lo, lo + args.minibatch)
            mu, value = net_forward(obs[sl])
            nlp = 0.5 * (((act[sl] - mu) / torch.exp(logstd)) ** 2).sum(dim=1) + 0.5 * math.log(2.0 * math.pi) * A + logstd.sum()
            ratio = torch.exp(old_nlp[sl] - nlp)
            a_loss = torch.max(-adv[sl] * ratio, -adv[sl] * torch.clamp(ratio, 1.0 - E_CLIP, 1.0 + E_CLIP))
            v_clip = old_v[sl] + torch.clamp(value - old_v[sl], -E_CLIP, E_CLIP)
            c_loss = torch.max((value - ret[sl]) ** 2, (v_clip - ret[sl]) ** 2)
            b_loss = (torch.clamp(mu - 1.1, min=0.0) ** 2 + torch.clamp(-1.1 - mu, min=0.0) ** 2).sum(dim=1)
            loss = a_loss.mean() + 0.5 * CRITIC_COEF * c_loss.mean() + BOUNDS_COEF * b_loss.mean()
            opt.zero_grad(set_to_none=True)
            (loss * scale).backward()
            if use_dist:          # what the native learner has not already averaged inside its backward: everything (torch learner) or the log-std alone
                rest = [p.grad for p in params if p.grad is not None] if args.learner == "torch" else [logstd.grad]
                flat = torch.cat([g.reshape(-1) for g in rest])
                dist.all_reduce(flat)
                flat.div_(world)
                off = 0
                for g in rest:
                    g.copy_(flat[off:off + g.numel()].view_as(g)); off += g.numel()
            for p in params:
                if p.grad is not None:
                    p.grad.div_(scale)
            torch.nn.utils.clip_grad_norm_(params, GRAD_NORM)                                        # truncate_grads
            opt.step()
            steps += 1
    native.eval()
    if args.learner == "native":          # the rollout reads the fp16 operand images directly: recast them after the last optimizer step
        native.learner.sync_weights()
        native._seen = native._versions()
    return steps, float(loss.detach())


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


native.eval()
# a run that has been training for a while: the input statistics have seen many rows, so a minibatch moves them little (with the fresh count of 1 the first
# minibatches shift the normalisation — and with it mu — so far that the probability ratio overflows: rl_games would do the same on its first epochs)
for _ in range(3):
    col.collect().next_horizon()
    for t in range(H):
        native.learner.rms.update(col.obs[t])
learn()                                                       # warm-up: allocations, the native buffers at both row counts
roll_s = learn_s = 0.0
for _ in range(args.epochs):
    dt, _ = timed(lambda: col.collect().next_horizon())
    roll_s += dt
    col.sigma.copy_(torch.exp(logstd.detach()))
    dt, (steps, last_loss) = timed(learn)
    learn_s += dt
roll_ms, learn_ms = roll_s / args.epochs * 1e3, learn_s / args.epochs * 1e3
if use_dist:
    t = torch.tensor([roll_ms, learn_ms], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    roll_ms, learn_ms = float(t[0]), float(t[1])
if rank == 0:
  print(json.dumps({
    "what": "one PPO epoch of BASELINE config 5's per-GPU slice (27-dof task, rl_games a2c_continuous settings of cfg/train/HumanoidPingpongTiltG1PPO.yaml), context only",
    "learner": args.learner, "num_envs": n, "horizon": H, "minibatch_rows": args.minibatch, "mini_epochs": MINI_EPOCHS, "minibatch_steps_per_epoch": steps,
    "ms_rollout_per_epoch": roll_ms, "ms_learning_per_epoch": learn_ms, "ms_per_minibatch_step": learn_ms / steps,
    "ranks": world, "gradient_all_reduce": ("RCCL, one bucket per layer beside the backward" if args.learner == "native" else "RCCL, one flat bucket after the backward") if use_dist else "none (one rank)",
    "env_steps_per_s_end_to_end": world * n * H / ((roll_ms + learn_ms) * 1e-3), "last_loss": last_loss,
    "finite": bool(all(torch.isfinite(p).all() for p in params))}))
if use_dist:
    dist.barrier()
    dist.destroy_process_group()
