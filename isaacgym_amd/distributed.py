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
`GradientBuckets` is the learners' side of that mode: the gradient all-reduce, one bucket per layer, issued while the
backward of the layers below is still running.
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


def gather_rollout(local, group=None, force=False, pad_to=None):
    """All-gather a per-rank tensor whose dim 0 is the env dim into the global tensor, in global env order.
    Ranks may own different counts (shard_range); ragged shards are padded to the largest and trimmed.
    force: issue the collective even in a one-rank group (the RCCL path on a one-GPU box); pad_to: pad every shard to at least
    this many rows before the collective (with one rank: exercises the ragged path — pad, gather, trim)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return local
    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    counts[dist.get_rank(group)] = local.shape[0]
    dist.all_reduce(counts, group=group)
    counts = counts.tolist()
    m = max(max(counts), int(pad_to or 0))
    padded = local
    if local.shape[0] < m:
        padded = torch.cat([local, local.new_zeros((m - local.shape[0],) + tuple(local.shape[1:]))])
    out = local.new_empty((world * m,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)])


class RolloutGather:
    """The north star's "episodic RCCL gather of obs/reward" for a central learner, without a host synchronisation and
    without making the step stream wait: shard sizes are exchanged ONCE (they are static: shard_range), every later call is one
    asynchronous all_gather_into_tensor per tensor into a preallocated rank-major buffer, joined only when its slot comes round
    again (`depth` rotating slots, like AsyncHorizonStats) or in `result()`.

    A tensor's env dim is `env_dim` (0 for per-step tensors [n, ...], 1 for horizon-major ones [H, n, ...]).  Ragged shards: the
    send buffer is the slot's own padded staging tensor [.., m, ..] (m = the largest shard); equal shards are sent in place.
    `result(slot, i)` returns tensor i in GLOBAL env order ([.., sum(counts), ..]): a view of the receive buffer when the layout
    allows (per-step tensors of equal shards), else one gather copy."""

    def __init__(self, count, device, group=None, depth=2, force=False, pad_to=None):
        self.group, self.depth, self.device = group, int(depth), torch.device(device)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        c = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        c[self.rank] = int(count)
        if self.active:
            dist.all_reduce(c, group=group)
        self.counts = [int(x) for x in c.tolist()]           # once, at construction: the only host read of this class
        self.m = max(max(self.counts), int(pad_to or 0))
        self.ragged = any(x != self.m for x in self.counts)
        self.slots = [None] * self.depth                     # per slot: list of (recv, staging or None, env_dim)
        self.works = [[] for _ in range(self.depth)]
        self.bytes_sent = 0

    def _buffers(self, slot, tensors, env_dims):
        if self.slots[slot] is None:
            bufs = []
            for t, ed in zip(tensors, env_dims):
                shape = list(t.shape)
                shape[ed] = self.m
                recv = torch.empty([self.world] + shape, dtype=t.dtype, device=self.device)
                stage = torch.zeros(shape, dtype=t.dtype, device=self.device) if t.shape[ed] != self.m else None
                bufs.append((recv, stage, ed))
            self.slots[slot] = bufs
        return self.slots[slot]

    def push(self, slot, tensors, env_dims=None):
        """Issue the gathers of `tensors` (this rank's shard of each) into slot `slot`; returns immediately."""
        env_dims = [0] * len(tensors) if env_dims is None else list(env_dims)
        self.wait(slot)
        for t, (recv, stage, ed) in zip(tensors, self._buffers(slot, tensors, env_dims)):
            send = t
            if stage is not None:                            # ragged: pad this rank's shard to the largest
                stage.narrow(ed, 0, t.shape[ed]).copy_(t)
                send = stage
            send = send if send.is_contiguous() else send.contiguous()
            self.bytes_sent += send.numel() * send.element_size()
            if self.active:
                self.works[slot].append(dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group, async_op=True))
            else:
                recv[0].copy_(send)

    def wait(self, slot):
        for w in self.works[slot]:
            w.wait()
        self.works[slot] = []

    def result(self, slot, i):
        """Tensor i of slot `slot` in global env order."""
        self.wait(slot)
        recv, _, ed = self.slots[slot][i]                    # [world, .., m, ..]
        parts = [recv[r].narrow(ed, 0, self.counts[r]) for r in range(self.world)]
        if self.world == 1:
            return parts[0]
        if ed == 0 and not self.ragged:
            return recv.reshape((self.world * self.m,) + tuple(recv.shape[2:]))
        return torch.cat(parts, dim=ed)


class GradientBuckets:
    """Data-parallel learners (the reference's multi-GPU mode: one env shard + one learner per rank, `train.py:117-120`; rl_games averages the
    gradients over the ranks before every optimizer step).  A callable for `NativeMLPLearner.backward(on_grads=...)`: each layer's gradient tensors
    are all-reduced as soon as the backward has ENQUEUED the kernels that produce them — asynchronously, so the collective of layer l runs on RCCL's
    stream beside the dX / dW kernels of the layers below it (a layer's dW of both networks is 0.5–25 MB: a bucket by itself; xGMI is point to point,
    large buckets keep its per-link rings busy).  `wait()` joins them (the compute stream then waits for the collectives) and turns the sums into means;
    call it before the optimizer step.  force: issue the collectives even in a one-rank group (the RCCL path on a one-GPU box)."""

    def __init__(self, group=None, force=False):
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.active else 1
        self.works, self.tensors, self.names, self.bytes = [], [], [], 0
        self._joined = True                     # wait() has run since the last bucket: the next bucket starts a new backward

    def __call__(self, name, tensors):
        """Not for `backward(accumulate=True)`: the gradient buffers then hold sums that earlier backward passes already reduced, and reducing
        them again would count those once more — accumulate locally (on_grads=None) and pass this callable to the LAST micro-batch's backward only
        (NativeMLPLearner.backward refuses the combination)."""
        if self._joined:                        # `names` / `bytes` describe ONE backward: they start over here instead of growing for ever
            self.names, self.bytes, self._joined = [], 0, False
        self.names.append(name)
        for t in tensors:
            self.tensors.append(t)
            self.bytes += t.numel() * t.element_size()
            if self.active:
                self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.works:
            w.wait()
        if self.world > 1:
            for t in self.tensors:
                t.div_(self.world)
        self.works, self.tensors = [], []
        self._joined = True
