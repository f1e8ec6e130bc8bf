"""`python bench.py --gpus N` as the driver may invoke it: with WORLD_SIZE unset the script starts its own rank processes.
CPU rehearsal of that entry (gloo, no env): the launch, the rendezvous on 127.0.0.1, rank 0's single JSON line, the exit code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_launches_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "7", "--warmup", "3", "--rehearse", "--dist-backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                    # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["warmup"] == 3
    assert out["max_over_ranks"] == 2.0                 # the all-reduce(MAX) saw both ranks


def test_bench_parent_reports_a_failed_rank_and_does_not_hang():
    r = _run(["--gpus", "2", "--rehearse"], {"PPENV_BENCH_REHEARSE_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode == 3
    assert r.stdout.strip() == ""


def test_bench_under_an_external_launcher_is_one_rank_of_it():
    # what `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` sets: no self-launch
    r = _run(["--gpus", "1", "--rehearse"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout)["n_gpus"] == 1


@pytest.mark.gpu
def test_bench_gpus_2_real_entry_on_one_gpu_gloo():
    """The real entry with two ranks sharing cuda:0 (gloo for the three-scalar all-reduce: RCCL refuses two ranks on one GPU)."""
    r = _run(["--gpus", "2", "--dist-backend", "gloo", "--num-envs", "2048", "--steps", "64", "--warmup", "32", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["global_envs"] == 4096 and out["value"] > 0
    assert out["roofline"]["avg_kernel_us"] > 0 and "cpu_baseline" not in out


@pytest.mark.gpu
def test_bench_rccl_path_with_one_rank():
    """RCCL on the hardware at hand: one rank, `--force-dist` — the process group is created with backend nccl (= RCCL) and device_id,
    every horizon's statistics go through an asynchronous float64 all-reduce beside the graph replays, the timing goes through the
    barrier and the max-reduce.  (More than one rank per GPU is refused by RCCL, and multi-GPU runs are the driver's.)"""
    r = _run(["--gpus", "1", "--force-dist", "--num-envs", "4096", "--steps", "128", "--warmup", "32", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["episode_stats"]["mean_progress"] > 0


def _last_json(r):
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rew", "obs"])
def test_bench_rccl_gather_with_one_rank(mode):
    """The north star's episodic gather of rewards / dones (and, `obs`, every step's observation rows) on the `nccl` backend (= RCCL)
    with the one rank a one-GPU box has: asynchronous all_gather_into_tensor through two rotating slots beside the graph replays,
    shards padded to a fake larger count so that the ragged path (pad, gather, trim) runs too."""
    out = _last_json(_run(["--gpus", "1", "--force-dist", "--gather", mode, "--gather-pad", "4160", "--num-envs", "4096", "--steps", "128",
                           "--warmup", "32", "--no-cpu-baseline"], timeout=900))
    g = out["gather"]
    assert g["mode"] == mode and g["backend"] == "nccl" and g["ranks"] == 1 and g["shard_rows"] == [4096] and g["padded_rows"] == 4160 and g["ragged"]
    assert g["horizon_gathers_issued"] >= (128 + 32) // 32 and g["collectives"]["horizon_rew_done"]["us_per_call_blocking"] > 0
    assert g["collectives"]["horizon_rew_done"]["bytes_sent_per_rank_per_call"] == 32 * 4096 * (4 + 8)
    if mode == "obs":
        assert g["step_gathers_issued"] >= 128 and g["collectives"]["step_obs"]["bytes_sent_per_rank_per_call"] == 4096 * 80 * 4
        assert out["config"]["launch"] == "eager"
    assert out["value"] > 0 and "configs" not in out


@pytest.mark.gpu
def test_bench_rollout_workload_with_rccl_gather():
    """BASELINE configs[4]'s per-GPU slice as a bench.py workload (27-dof step + native policy forward), through the same rank
    plumbing as the step: process group on nccl with one rank, rewards / dones gathered per horizon."""
    out = _last_json(_run(["--gpus", "1", "--workload", "rollout", "--force-dist", "--gather", "rew", "--num-envs", "1024", "--steps", "64",
                           "--warmup", "32", "--no-cpu-baseline"], timeout=900))
    assert out["config"]["variant"] == "TA" and out["config"]["kind"] == "rollout" and out["value"] > 0
    assert out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1 and out["gather"]["ranks"] == 1


@pytest.mark.gpu
def test_bench_rollout_two_ranks_on_one_gpu_gloo():
    out = _last_json(_run(["--gpus", "2", "--workload", "rollout", "--dist-backend", "gloo", "--gather", "rew", "--num-envs", "512", "--steps", "64",
                           "--warmup", "32", "--no-cpu-baseline"], timeout=900))
    assert out["n_gpus"] == 2 and out["config"]["global_envs"] == 1024 and out["gather"]["ranks"] == 2 and out["gather"]["shard_rows"] == [512, 512]


@pytest.mark.gpu
def test_bench_secondary_config_row():
    """One of the `configs` rows on its own (--only-config): BASELINE configs[1] at its named size, with roofline and cpu_baseline."""
    r = _run(["--only-config", "c2_TT_4096"], timeout=900)
    row = _last_json(r)["configs"][0]
    assert row["variant"] == "TT" and row["num_envs"] == 4096 and row["avg_kernel_us"] > 0
    assert row["roofline"]["bound"] == "hbm" and row["roofline"]["algorithmic_bytes_per_launch"] == 608 * 4096 and row["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_rollout_gather_values_on_rccl_with_one_rank(tmp_path):
    """RolloutGather itself on the nccl backend, values checked: equal and padded shards, horizon-major and per-step tensors."""
    code = '''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from isaacgym_amd import distributed as D
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for pad in (0, 300):
    g = D.RolloutGather(257, dev, depth=2, force=True, pad_to=pad)
    assert g.active and g.counts == [257] and g.m == max(257, pad)
    for hz in range(5):
        rew = torch.randn(32, 257, device=dev); done = torch.randint(0, 2, (32, 257), device=dev); obs = torch.randn(257, 80, device=dev)
        g.push(hz %% 2, [rew, done, obs], env_dims=[1, 1, 0])
        for i, t in enumerate((rew, done, obs)):
            assert torch.equal(g.result(hz %% 2, i), t), (pad, hz, i)
    x = torch.randn(257, 7, device=dev)
    assert torch.equal(D.gather_rollout(x, force=True, pad_to=pad), x)
dist.barrier(); dist.destroy_process_group(); print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]
