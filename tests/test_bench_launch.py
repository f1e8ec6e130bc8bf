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
