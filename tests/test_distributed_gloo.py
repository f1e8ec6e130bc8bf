"""The N>1 path on CPU: world_size-2 gloo processes, each owning an env shard (stepped by the CPU oracle here,
since no GPU exists in this container), exchanging only what isaacgym_amd.distributed exchanges."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_GLOBAL, STEPS, SEED = 101, 60, 5     # 101: ragged split (51 + 50)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _actions(step):
    return np.random.default_rng(1000 + step).uniform(-1, 1, (N_GLOBAL, 7)).astype(np.float32)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from isaacgym_amd import distributed as D
    from isaacgym_amd import scene
    from oracle import binding as ob
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert D.rank_info() == (rank, rank, world)
    off, cnt = D.shard_range(N_GLOBAL, rank, world)
    env = ob.OracleEnv(scene.build_config("TT", num_envs=cnt, seed=SEED, env_id_offset=off))
    stats = []
    for t in range(STEPS):
        env.step(_actions(t)[off:off + cnt])
        if (t + 1) % 20 == 0:
            stats.append(D.horizon_stats(torch.from_numpy(env.rew_buf.copy()), torch.from_numpy(env.progress_buf.copy()),
                                         torch.from_numpy(env.episode.astype(np.int64))).numpy())
    obs = D.gather_rollout(torch.from_numpy(env.obs_buf.copy()))
    rew = D.gather_rollout(torch.from_numpy(env.rew_buf.copy()))
    reset = D.gather_rollout(torch.from_numpy(env.reset_buf.copy()))
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), obs=obs.numpy(), rew=rew.numpy(), reset=reset.numpy(), stats=np.stack(stats))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_reproduces_the_single_process_run(oracle_lib, tmp_path):
    from isaacgym_amd import distributed as D
    from isaacgym_amd import scene
    assert D.shard_range(101, 0, 2) == (0, 51) and D.shard_range(101, 1, 2) == (51, 50)
    assert sum(D.shard_range(65536, r, 8)[1] for r in range(8)) == 65536 and D.shard_range(65536, 3, 8) == (3 * 8192, 8192)
    with pytest.raises(ValueError):
        D.shard_range(10, 2, 2)
    oracle_lib.build()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npz")
    # single process, all envs in one handle
    env = oracle_lib.OracleEnv(scene.build_config("TT", num_envs=N_GLOBAL, seed=SEED))
    stats = []
    for t in range(STEPS):
        env.step(_actions(t))
        if (t + 1) % 20 == 0:
            stats.append([env.rew_buf.astype(np.float64).mean(), env.progress_buf.astype(np.float64).mean(), float(env.episode.sum())])
    np.testing.assert_array_equal(got["obs"], env.obs_buf)       # env i's trajectory does not depend on the sharding
    np.testing.assert_array_equal(got["rew"], env.rew_buf)
    np.testing.assert_array_equal(got["reset"], env.reset_buf)
    np.testing.assert_allclose(got["stats"], np.array(stats), rtol=1e-12, atol=1e-9)
    assert got["obs"].shape == (N_GLOBAL, 80)


class _FakeEnv:
    """Stands in for PPEnv in the CPU test of AsyncHorizonStats: reduce_stats over oracle arrays."""

    def __init__(self, oracle_env):
        self.o, self.device = oracle_env, torch.device("cpu")

    def reduce_stats(self, out=None):
        out[0], out[1] = float(self.o.rew_buf.astype(np.float64).sum()), float(self.o.progress_buf.sum())
        out[2], out[3] = float(self.o.episode.sum()), float(self.o.num_envs)
        return out


def _async_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from isaacgym_amd import distributed as D
    from isaacgym_amd import scene
    from oracle import binding as ob
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = D.shard_range(N_GLOBAL, rank, world)
    env = ob.OracleEnv(scene.build_config("TT", num_envs=cnt, seed=SEED, env_id_offset=off))
    stats = D.AsyncHorizonStats(_FakeEnv(env), depth=2)
    for t in range(STEPS):
        env.step(_actions(t)[off:off + cnt])
        if (t + 1) % 10 == 0:
            stats.push()          # 6 horizons through 2 rotating buffers
    if rank == 0:
        np.save(os.path.join(out_dir, "async_stats.npy"), stats.latest().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_async_horizon_stats_two_ranks(oracle_lib, tmp_path):
    from isaacgym_amd import scene
    oracle_lib.build()
    mp.spawn(_async_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "async_stats.npy")
    env = oracle_lib.OracleEnv(scene.build_config("TT", num_envs=N_GLOBAL, seed=SEED))
    for t in range(STEPS):
        env.step(_actions(t))
    want = [env.rew_buf.astype(np.float64).mean(), env.progress_buf.astype(np.float64).mean(), float(env.episode.sum())]
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-9)


def test_single_process_helpers_are_identity():
    from isaacgym_amd import distributed as D
    x = torch.arange(12.0).view(6, 2)
    assert D.gather_rollout(x) is x
    s = D.horizon_stats(torch.tensor([1.0, 3.0]), torch.tensor([4, 6]), torch.tensor([2, 5]))
    assert s.tolist() == [2.0, 5.0, 7.0]


# ---- BASELINE config 4 (4-actor, sharded, obs / reward gather) and config 5 (27-dof) on two gloo ranks --------------
N4, STEPS4 = 37, 45      # 37 envs = 74 agent rows: ragged split 19 + 18 envs


def _actions4(step):
    return np.random.default_rng(2000 + step).uniform(-1, 1, (2 * N4, 7)).astype(np.float32)


def _worker_t4_ta(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from isaacgym_amd import distributed as D
    from isaacgym_amd import scene
    from oracle import binding as ob
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = D.shard_range(N4, rank, world)
    env = ob.OracleEnv(scene.build_config("T4", num_envs=cnt, seed=SEED, env_id_offset=off))
    for t in range(STEPS4):
        env.step(_actions4(t)[2 * off:2 * (off + cnt)])      # agent rows 2e, 2e + 1 travel with env e
    obs = D.gather_rollout(torch.from_numpy(env.obs_buf.copy()))
    rew = D.gather_rollout(torch.from_numpy(env.rew_buf.copy()))
    # 27-dof task: the reset draws are keyed by the global env id, so a shard lays out the same initial balls
    p = scene.build_ta_params(cnt, seed=SEED, env_id_offset=off)
    draws = D.gather_rollout(scene.ta_reset_draws(p, torch.arange(cnt), torch.zeros(cnt, dtype=torch.int64)))
    if rank == 0:
        np.savez(os.path.join(out_dir, "t4.npz"), obs=obs.numpy(), rew=rew.numpy(), draws=draws.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_of_the_two_agent_and_27dof_tasks(oracle_lib, tmp_path):
    from isaacgym_amd import scene
    oracle_lib.build()
    mp.spawn(_worker_t4_ta, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "t4.npz")
    env = oracle_lib.OracleEnv(scene.build_config("T4", num_envs=N4, seed=SEED))
    for t in range(STEPS4):
        env.step(_actions4(t))
    assert got["obs"].shape == (2 * N4, 80)
    np.testing.assert_array_equal(got["obs"], env.obs_buf)
    np.testing.assert_array_equal(got["rew"], env.rew_buf)
    p = scene.build_ta_params(N4, seed=SEED)
    want = scene.ta_reset_draws(p, torch.arange(N4), torch.zeros(N4, dtype=torch.int64)).numpy()
    np.testing.assert_array_equal(got["draws"], want)
    assert (want[:, 0] >= -0.5).all() and (want[:, 0] <= 0.1).all() and (want[:, 2] < -4.0).all()   # TA:133-134, serve towards the humanoid


# ---- the episodic gather for a central learner (RolloutGather): horizon-major rewards / dones, per-step observation rows ------
def _gather_worker(rank, world, port, out_dir, n_global):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from isaacgym_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = D.shard_range(n_global, rank, world)
    H = 4
    g = D.RolloutGather(cnt, "cpu", depth=2)
    assert g.counts == [D.shard_range(n_global, r, world)[1] for r in range(world)] and g.ragged == (n_global % world != 0)
    ids = torch.arange(off, off + cnt)
    got = {}
    for hz in range(3):                                   # three horizons through two slots
        slot = hz % 2
        rew = (1000.0 * hz + 10.0 * torch.arange(H)[:, None] + ids[None, :].float() / 1000)          # [H, cnt]
        done = (ids[None, :] + torch.arange(H)[:, None] + hz).to(torch.int64)                        # [H, cnt]
        obs = (ids[:, None].float() + torch.arange(5)[None, :] / 10 + hz)                            # [cnt, 5]
        g.push(slot, [rew, done, obs], env_dims=[1, 1, 0])
        got[hz] = [g.result(slot, i).clone() for i in range(3)]
    if rank == 0:
        torch.save(got, os.path.join(out_dir, f"gather_{n_global}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [12, 13])            # equal shards (views of the receive buffer) and ragged ones (7 + 6)
def test_rollout_gather_two_ranks(tmp_path, n_global):
    mp.spawn(_gather_worker, args=(2, _free_port(), str(tmp_path), n_global), nprocs=2, join=True)
    got = torch.load(tmp_path / f"gather_{n_global}.pt")
    ids, H = torch.arange(n_global), 4
    for hz in range(3):
        rew, done, obs = got[hz]
        assert rew.shape == (H, n_global) and done.shape == (H, n_global) and obs.shape == (n_global, 5) and done.dtype == torch.int64
        torch.testing.assert_close(rew, 1000.0 * hz + 10.0 * torch.arange(H)[:, None] + ids[None, :].float() / 1000, rtol=0, atol=0)
        assert torch.equal(done, (ids[None, :] + torch.arange(H)[:, None] + hz).to(torch.int64))
        torch.testing.assert_close(obs, ids[:, None].float() + torch.arange(5)[None, :] / 10 + hz, rtol=0, atol=0)


def test_rollout_gather_single_process_is_a_local_copy_and_pads():
    from isaacgym_amd import distributed as D
    g = D.RolloutGather(5, "cpu", depth=2, pad_to=8)      # no process group: the padded staging path alone
    assert g.counts == [5] and g.m == 8 and g.ragged
    x = torch.arange(15.0).view(3, 5)
    g.push(1, [x], env_dims=[1])
    assert torch.equal(g.result(1, 0), x)
    assert g.bytes_sent == 3 * 8 * 4


def _grad_bucket_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from isaacgym_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = D.GradientBuckets()
    assert b.active and b.world == world
    layers = {"heads": [torch.full((8, 64), float(rank + 1)), torch.full((8,), 10.0 * (rank + 1))], "layer2": [torch.arange(12.0).view(2, 2, 3) * (rank + 1), torch.ones(2, 2) * rank]}
    for name, ts in layers.items():          # the learner's backward calls it layer by layer; the collectives are in flight until wait()
        b(name, ts)
    assert b.names == ["heads", "layer2"] and b.bytes == (8 * 64 + 8 + 12 + 4) * 4
    b.wait()
    b("heads", [torch.zeros(3)])             # the next backward: names / bytes describe one backward, they do not grow for ever
    assert b.names == ["heads"] and b.bytes == 12
    b.wait()
    if rank == 0:
        torch.save(layers, os.path.join(out_dir, "buckets.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_average_over_two_ranks(tmp_path):
    """distributed.GradientBuckets (the data-parallel learners' per-layer gradient all-reduce): sums over the ranks, then the mean, in place."""
    mp.spawn(_grad_bucket_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "buckets.pt")
    assert torch.equal(got["heads"][0], torch.full((8, 64), 1.5)) and torch.equal(got["heads"][1], torch.full((8,), 15.0))
    assert torch.equal(got["layer2"][0], torch.arange(12.0).view(2, 2, 3) * 1.5) and torch.equal(got["layer2"][1], torch.full((2, 2), 0.5))


def test_gradient_buckets_without_a_process_group_leave_the_gradients_alone():
    from isaacgym_amd import distributed as D
    b = D.GradientBuckets()
    t = torch.arange(6.0)
    b("layer1", [t])
    b.wait()
    assert not b.active and torch.equal(t, torch.arange(6.0)) and b.names == ["layer1"]
