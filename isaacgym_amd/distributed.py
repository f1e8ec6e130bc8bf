"""Multi-GPU sharding of the env set: one process and one native handle per GPU.

The reference's multi-GPU mode (pingpong_note.txt:163: `torchrun --nproc_per_node=7 train.py
multi_gpu=True`) gives every rank its own simulator, env shard and learner; the only traffic between
ranks is rl_games' gradient all-reduce.  The env step itself has no cross-env term (every reward /
obs op is elementwise over dim 0, TT:1144-1265), so the data path shards with NO collective:
contiguous blocks of env ids, `num_envs_per_rank` each.  Trajectories are keyed by the GLOBAL env id
(the RNG counter is (seed, global id, episode)), so a sharded run reproduces the single-handle run
env for env.

What is exchanged, once per horizon, is what the reference prints every 40 steps (TT:763-766) — mean
reward and mean progress — plus the episode count: three scalars, one all-reduce over RCCL
(backend "nccl" on ROCm) or gloo in the CPU tests.  `gather_rollout` is the optional
obs/reward/done all-gather for a single central learner; DESIGN.md explains why it cannot be the
default (the env emits more obs bytes per second than one GPU's xGMI links can absorb).
"""
import os

import torch
import torch.distributed as dist


def shard_range(num_envs_global, rank, world_size):
    """Contiguous block of global env ids owned by `rank`: (offset, count).  Remainder goes to the low ranks."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, rem = divmod(int(num_envs_global), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def rank_info():
    """(rank, local_rank, world_size) from the torch.distributed launcher's environment (reference train.py:117)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def env_horizon_stats(env, group=None):
    """`horizon_stats` of a native PPEnv: one reduction launch (ppenv_reduce_stats) + one 4-double all-reduce."""
    s = env.reduce_stats()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return torch.stack([s[0] / s[3], s[1] / s[3], s[2]])


class AsyncHorizonStats:
    """The same statistics without ever making the step stream wait for the collective: the all-reduce of
    horizon k is issued asynchronously into one of `depth` rotating buffers and only joined when its buffer
    comes round again (or in `latest()`).  A synchronous all-reduce would serialise ~30 us of RCCL latency into
    every 32-step horizon of a ~16 us step."""

    def __init__(self, env, depth=4, group=None, force=False):
        """force: issue the all-reduce even in a one-rank group (bench.py --force-dist: the RCCL path on a one-GPU box)."""
        self.env, self.group, self.depth = env, group, depth
        self.bufs = [torch.zeros(4, dtype=torch.float64, device=env.device) for _ in range(depth)]
        self.works = [None] * depth
        self.k = -1
        self.multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)

    def push(self):
        self.k += 1
        slot = self.k % self.depth
        if self.works[slot] is not None:
            self.works[slot].wait()
            self.works[slot] = None
        self.env.reduce_stats(out=self.bufs[slot])
        if self.multi:
            self.works[slot] = dist.all_reduce(self.bufs[slot], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def latest(self):
        """[mean reward, mean progress, finished episodes] of the most recent horizon (joins outstanding work)."""
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = [None] * self.depth
        s = self.bufs[self.k % self.depth] if self.k >= 0 else self.bufs[0]
        return torch.stack([s[0] / s[3].clamp(min=1), s[1] / s[3].clamp(min=1), s[2]])


def horizon_stats(rew_buf, progress_buf, episode, group=None):
    """[mean reward, mean progress, finished episodes] over ALL ranks' envs (tensor of 3 float64)."""
    n = torch.tensor(float(rew_buf.numel()), dtype=torch.float64, device=rew_buf.device)
    s = torch.stack([rew_buf.double().sum(), progress_buf.double().sum(), episode.double().sum(), n])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return torch.stack([s[0] / s[3], s[1] / s[3], s[2]])


def gather_rollout(local, group=None):
    """All-gather a per-rank tensor whose dim 0 is the env dim into the global tensor, in global env order.
    Ranks may own different counts (shard_range); ragged shards are padded to the largest and trimmed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[dist.get_rank(group)] = local.shape[0]
    dist.all_reduce(counts, group=group)
    counts = counts.tolist()
    m = max(counts)
    padded = local
    if local.shape[0] < m:
        padded = torch.cat([local, local.new_zeros((m - local.shape[0],) + tuple(local.shape[1:]))])
    out = local.new_empty((world * m,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)])
