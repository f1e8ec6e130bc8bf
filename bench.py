#!/usr/bin/env python3
"""bench.py — env-steps/s of the fused HumanoidPingpong VecTask step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one VecTask.step() of every env this rank owns = one launch of the fused kernel
(action->PD target, 2 physics substeps, reward, masked reset, observations).  Workload at N=1:
BASELINE.json configs[2], "humanoid_pingpong 3-actor tilt, num_envs=16384 on 1 MI355X" (the
config the metric "env-steps/sec at N_envs=16384" is quoted on).  For N>1 every rank owns its own
16384 envs on its own GPU (weak scaling: the reference's multi-GPU mode is one env shard + one
learner per rank, pingpong_note.txt:163); env ids are offset per rank so trajectories do not
depend on the sharding.  The data path has no collective; once per PPO horizon (32 steps,
cfg/train/HumanoidPingpongTiltG1PPO.yaml:73) the ranks all-reduce the three scalars the
reference prints (mean reward, mean progress — TT:763-766 — and the episode count) over RCCL.

Inputs are synthetic and resident in HBM before the timed region: a pool of U(-1,1) action
tensors from torch.Generator(seed 0).  Timing: W warm-up steps, barrier + synchronize, exactly K
steps, synchronize + barrier; the max over ranks is used.  Rank 0 prints ONE JSON line.

The steps of one PPO horizon (32 launches of the step kernel on the pool's action tensors) are
captured once into a HIP graph and replayed: the launches are the same kernels on the same
stream in the same order, the graph only removes the per-launch submission from the loop
(rocprofv3's kernel trace shows the per-kernel duration drop from 12.6 to 11.8 us).  K steps =
K // 32 replays + one replay of a (K % 32)-step graph; --no-graph launches every step eagerly.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NUM_ENVS = 16384
VARIANT = "TT"
WORKLOAD_NAMES = {"TT": "3-actor tilt (HumanoidPingpongTiltG1)", "TN": "3-actor tilt, no early stop (HumanoidPingpongTiltNoEarlyStopG1)",
                  "T3": "3-actor (HumanoidPingpongG1)", "T4": "4-actor tilt (Humanoid12PingpongTiltG1: two humanoids, two agent rows per env)",
                  "TA": "3-actor all-dof (HumanoidPingpongTiltNESSparse27DOF: free-floating 27-dof humanoid, 313 observations)"}
HORIZON = 32
# SURVEY.md §8(d) "minimal algorithmic bytes / env-step": 7-DoF variants R 156 + W 452 = 608; 4-actor ("~1.1 KB"): R actions 56 +
# q,qd 112 + ball 52 + progress/flags/prev-vx 16 + rng 8 = 244, W q,qd 112 + ball 52 + misc 24 + obs 640 + rew 8 + reset 8 = 844
# 27-dof ("~2.0 KB"): R 444 + W 1596 (obs 1252)
ALGO_BYTES = {"TT": 608, "TN": 608, "T3": 608, "T4": 1088, "TA": 2040}
ALGO_BYTES_PER_ENV_STEP = ALGO_BYTES["TT"]
PREWARM = 512
ROOFLINE_WARM, ROOFLINE_LAUNCHES, ROOFLINE_REGIONS = 256, 480, 5   # multiples of HORIZON; fixed, whatever --steps is
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def usable_cores():
    """Host cores this process may really use: the cgroup CPU quota if there is one, else the affinity mask,
    capped at 16 (a one-GPU box's CPU share); oversubscribing OpenMP threads only slows the baseline down."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, 16))


def cpu_baseline(num_envs, target_seconds=12.0, variant=VARIANT):
    """The CPU oracle (oracle/ppenv_oracle.c, kind 'port') timed on this box's host cores on a bounded
    sample of the same workload.  Reported, not targeted."""
    import numpy as np
    from isaacgym_amd import scene
    from oracle import binding as ob
    ob.build()
    cores = usable_cores()
    if variant == "TA":
        return cpu_baseline_ta(num_envs, cores, target_seconds)
    env = ob.OracleEnv(scene.build_config(variant, num_envs=num_envs, seed=0), threads=cores)
    rng = np.random.default_rng(0)
    actions = [rng.uniform(-1, 1, (num_envs * env.num_agents, 7)).astype(np.float32) for _ in range(4)]
    env.step(actions[0])   # warm-up
    t0 = time.perf_counter()
    steps = 0
    while True:
        env.step(actions[steps % 4])
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or steps >= 2000:
            break
    return {"value": num_envs * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of the {variant} variant at num_envs={num_envs}, OpenMP over envs, {dt:.1f} s"}


def cpu_baseline_rows(cores, seconds=2.5):
    """The rows BASELINE.md §3 promises next to the main cpu_baseline object, each on a bounded sample (`seconds` of CPU work):
    the oracle (fp64 Newton-Euler + dense solve — a different algorithm and precision from the kernel) at BASELINE
    config 1's shape (N = 64, T3 semantics: the runnable stand-in for tasks/humanoid_pingpong.py) and at N = 16384 TT,
    with 1 thread and with all cores; and the kernel's OWN arithmetic (csrc/ppenv_device.h compiled for the host: tests/csrc/host_shim.cpp,
    fp32 ABA, scalar, 1 thread) at N = 16384 TT."""
    import numpy as np
    from isaacgym_amd import scene
    from oracle import binding as ob
    rows = []

    def timed(step, n, label, kind_note, threads):
        rng = np.random.default_rng(0)
        acts = [rng.uniform(-1, 1, (n, 7)).astype(np.float32) for _ in range(4)]
        step(acts[0])
        t0, k = time.perf_counter(), 0
        while True:
            step(acts[k % 4])
            k += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or k >= 4000:
                break
        rows.append({"what": label, "value": n * k / dt, "unit": "env-steps/s", "cores": threads, "kind": "port", "arithmetic": kind_note,
                     "sample": f"{k} steps at num_envs={n}, {dt:.1f} s"})

    for variant, n in (("T3", 64), ("TT", 16384)):
        for threads in sorted({1, cores}):
            env = ob.OracleEnv(scene.build_config(variant, num_envs=n, seed=0), threads=threads)
            timed(env.step, n, f"oracle, {variant}, N={n}, {threads} thread(s)", "fp64 recursive Newton-Euler + dense solve (oracle/ppenv_oracle.c)", threads)
            env.close()
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import shim_binding as sb
        shim = sb.ShimEnv(scene.build_config("TT", num_envs=16384, seed=0))
        o = ob.OracleEnv(scene.build_config("TT", num_envs=16384, seed=0))
        shim.copy_state_from(o)
        o.close()
        timed(shim.step, 16384, "kernel arithmetic on the host, TT, N=16384, 1 thread", "fp32 articulated-body algorithm: csrc/ppenv_device.h via tests/csrc/host_shim.cpp", 1)
    except Exception as e:   # noqa: BLE001 - the host shim needs g++ on the box; the oracle rows stand without it
        rows.append({"what": "kernel arithmetic on the host", "error": f"{type(e).__name__}: {e}"})
    return rows


def cpu_baseline_ta(num_envs, cores, target_seconds):
    """27-dof variant: the oracle's rigid-body step (OpenMP over envs) + its post_physics_step (one thread)."""
    import numpy as np
    from isaacgym_amd import scene
    from oracle import binding as ob
    n = num_envs
    cfg, model, p = scene.build_ta_scene(n), scene.build_ta_model(), scene.build_ta_params(n)
    root = np.zeros((n, 3, 13), np.float32)
    for a in range(3):
        root[:, a, :7] = np.array(list(p.init_root[a]))
    root[:, 2, 7:10] = (-5.0, 0.0, 1.5)
    dof = np.zeros((n, 27, 2), np.float32)
    irb = ob.ta_forward_kinematics(model, root, dof)
    flags, episode, progress = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64)
    rng = np.random.default_rng(0)
    actions = [rng.uniform(-1, 1, (n, 27)).astype(np.float32) for _ in range(4)]
    t0 = time.perf_counter()
    steps = 0
    while True:
        rb, frc, pvx = ob.ta_simulate(cfg, model, actions[steps % 4], root, dof, threads=cores)
        ob.ta_post_physics_step(p, rb, irb, root, dof, frc, pvx, None, flags, episode, progress)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= target_seconds or steps >= 2000:
            break
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of the TA variant at num_envs={n}, OpenMP over envs in the rigid-body step, {dt:.1f} s"}


def pmc_traffic(num_envs, variant=VARIANT, kernel=None):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950).
    bench.py cannot collect counters itself; null when no profile of this kernel at this workload size is committed."""
    import glob
    best = None
    norm = lambda name: name.replace(" ", "").replace("pp::", "").rstrip(">")     # older profiles: no namespace, fewer template arguments (defaults added later)
    kernel = norm(kernel or kernel_name(variant))                                  # kernel: the name the handle reports (ppenv_step_kernel_name)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        d = json.load(open(f))
        k = norm(d.get("kernel", ""))
        if d.get("num_envs") == num_envs and k and (kernel.startswith(k) or k.startswith(kernel)):
            best = d["hbm_bytes_per_launch"]
    return best


VALU_PEAK_WAVE_INSTS_PER_S = 1024 * 2.4e9 / 4    # 256 CUs x 4 SIMDs, one wave-instruction per SIMD every 4 cycles, 2.4 GHz max clock (MI355X_MICROARCH.md)


def _pmc_profile(num_envs, variant):
    """(counters, file) of the newest committed profiles/*_pmc_summary.csv of this variant's step kernel taken at num_envs (the
    companion *_pmc_traffic.json names the size and the kernel), or (None, None)."""
    import csv
    import glob
    tag = {"T4": "_T4", "TA": "_TA"}.get(variant, "")
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r??_?{tag}_pmc_traffic.json"))):
        d = json.load(open(f))
        summ = f.replace("_pmc_traffic.json", "_pmc_summary.csv")
        if d.get("num_envs") != num_envs or not os.path.exists(summ):
            continue
        if variant not in ("T4", "TA") and ("_T4_" in f or "_TA_" in f):
            continue
        c = {row["counter"]: float(row["mean_per_dispatch"]) for row in csv.DictReader(open(summ))}
        best = (c, os.path.basename(summ))
    return best


def valu_roofline(variant, n, kernel_us):
    """The ceiling that actually binds these kernels ("dependency-depth bound", not HBM): VALU issue.
    achieved = SQ_INSTS_VALU per launch (wave-instructions, from the committed rocprofv3 PMC pass of this kernel at this size; bench.py
    cannot collect counters itself) / the kernel's launch duration measured live here; peak = 1024 SIMDs x one wave-instruction per
    4 cycles x 2.4 GHz.  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES of the same pass: the share of a wave's lifetime spent waiting (on a
    hand-off, a barrier, memory) — the dependency chain.  None when no profile of this kernel at this size is committed."""
    c, src = _pmc_profile(n, variant)
    if not c or "SQ_INSTS_VALU" not in c:
        return None
    achieved = c["SQ_INSTS_VALU"] / (kernel_us * 1e-6)
    out = {"bound": "valu_issue", "achieved": achieved / 1e9, "peak": VALU_PEAK_WAVE_INSTS_PER_S / 1e9, "unit": "G wave-instructions/s",
           "frac": achieved / VALU_PEAK_WAVE_INSTS_PER_S, "valu_insts_per_launch": c["SQ_INSTS_VALU"], "source": f"profiles/{src}"}
    if c.get("SQ_WAVE_CYCLES"):
        out["wait_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        out["valu_busy_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_WAVES"):
        out["waves_per_launch"] = c["SQ_WAVES"]
        out["simd_slots_occupied_frac"] = min(1.0, c["SQ_WAVES"] / 1024.0)
    return out


def kernel_name(variant):
    """FALLBACK only (pmc_traffic's file matching, and rows whose handle is gone): the schedule ppenv_create picks (isaacgym_amd/csrc/ppenv.hip): two waves per 64 envs unless the one-wave kernel is
    forced; three waves (two arm waves + the ball wave) for the 4-actor variant."""
    if variant == "T4":
        return "step_kernel_split<ModelG1, 2, 0, 1, false>" if os.environ.get("PPENV_STEP_KERNEL") == "split3" else "step_kernel_split<ModelG1, 2, 1, 1, false>"
    if variant == "TA":
        return {"lane": "ta_sim_kernel<true>", "quad": "ta_sim_quad_kernel<true, true>"}.get(os.environ.get("PPENV_TA_KERNEL"), "ta_chain_kernel<false>")
    split = os.environ.get("PPENV_STEP_KERNEL") != "fused"
    return "step_kernel_split<ModelG1, 1, 0, 1, false>" if split else "step_kernel<ModelG1, false>"   # <model, humanoids, who sweeps the geometry, ball waves, randomisation tables>


def launch_ranks(n, argv):
    """Start `n` rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), wait for all,
    print rank 0's stdout (the ONE JSON line) and return the worst return code.  Runs before anything imports torch."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread so that a chatty child can never block on a full pipe while we poll
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    worst = 0
    live = list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and worst == 0:
                worst = rc
        if worst != 0:                      # one rank failed: the others would wait at a barrier for ever
            for p in live:
                p.kill()                    # exactly the processes started above
            for p in live:
                p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = b"".join(chunks)
    for line in out0.decode(errors="replace").splitlines():      # the JSON line to stdout; library chatter (gloo prints there) to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    return worst


def rehearse(args):
    """The rank logic of main() without an env or a GPU (gloo): rendezvous, barrier, all-reduce(MAX) of a timing, one line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    fail = os.environ.get("PPENV_BENCH_REHEARSE_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        return 3                                   # a rank that dies before the rendezvous (test of the parent's clean-up)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "max_over_ranks": float(t.item()), "steps": args.steps, "warmup": args.warmup}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------------------------------------
# Workloads.  A workload owns one env (and, for the rollout, the policy network), a pool of synthetic inputs resident in HBM, and
# the HIP graphs of its launch sequence.  `step(s, slot, t)` enqueues ONE step; with `into` buffers (gather modes) step t of a
# horizon writes its rewards / dones (/ observations) straight into the horizon-major slices [slot, t] a gather then sends.
MFMA_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense fp16 / bf16
UNITS = [2048, 1536, 1024, 1024, 512, 512]     # cfg/train/HumanoidPingpongTiltG1PPO.yaml:29


class Workload:
    def __init__(self, variant, n, device, rank=0, world=1, rollout=False, into_depth=0, into_obs=False):
        import torch
        from isaacgym_amd import distributed as D
        from isaacgym_amd import scene
        self.torch, self.variant, self.n, self.device, self.rollout = torch, variant, n, device, rollout
        off, cnt = D.shard_range(n * world, rank, world)   # contiguous global env ids; trajectories do not depend on the split
        gen = torch.Generator(device=device).manual_seed(rank)
        if variant == "TA":           # the 27-dof task on Isaac-Gym-layout tensors (ppenv_ta_step)
            from isaacgym_amd.tensor_api import TAEnv
            with torch.cuda.device(device):
                self.env = TAEnv(cnt, device=device, seed=0, env_id_offset=off)
            self.rows, self.num_act, self.num_obs = cnt, 27, 313
        else:
            from isaacgym_amd.env import PPEnv
            self.env = PPEnv(scene.build_config(variant, num_envs=cnt, seed=0, device_id=device.index, env_id_offset=off), device=device)
            self.rows, self.num_act, self.num_obs = cnt * self.env.num_agents, 7, 80
        self.pool = [(torch.rand(self.rows, self.num_act, device=device, generator=gen) * 2 - 1).contiguous() for _ in range(8)]
        self.obs_buf = self.env.obs_buf
        self.kernel = self.env.sim.kernel_name if variant == "TA" else self.env.step_kernel_name    # asked of the handle, not guessed
        self.graphs = {}
        # horizon-major output slices for the gather modes: [depth, HORIZON, rows(, num_obs)]
        self.into = None
        if into_depth:
            z = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=device)
            self.into = {"rew": z(into_depth, HORIZON, self.rows), "done": z(into_depth, HORIZON, self.rows, dt=torch.int64),
                         "obs": z(into_depth, HORIZON, self.rows, self.num_obs) if into_obs else None}
        self.net = None
        if rollout:
            self._make_policy()

    def _make_policy(self):
        """The reference's a2c network (cfg/train/HumanoidPingpongTiltG1PPO.yaml:10-31,50-52): separate actor and critic MLPs
        [2048, 1536, 1024, 1024, 512, 512], ELU, fixed sigma, normalize_input, mixed precision; random-initialised (no checkpoint
        exists offline) on the hand-written MFMA forward (isaacgym_amd.policy.NativeMLP)."""
        torch = self.torch
        from isaacgym_amd.policy import NativeMLP

        def mlp(n_out):
            layers, d = [], self.num_obs
            for u in UNITS:
                layers += [torch.nn.Linear(d, u), torch.nn.ELU()]
                d = u
            layers.append(torch.nn.Linear(d, n_out))
            return [(m.weight, m.bias) for m in torch.nn.Sequential(*layers) if isinstance(m, torch.nn.Linear)]
        torch.manual_seed(0)
        actor, critic = mlp(self.num_act), mlp(1)
        dev = self.device
        self.net = NativeMLP(actor, critic, self.num_obs, dev, mean=torch.zeros(self.num_obs, device=dev),
                             var=torch.ones(self.num_obs, device=dev) - 1e-5, max_rows=self.rows)
        self.sigma = torch.ones(self.num_act, device=dev)          # fixed_sigma, const_initializer 0 -> exp(0)
        self.action_buf = torch.zeros(self.rows, self.num_act, device=dev)
        self.neglogp = torch.zeros(self.rows, device=dev)
        self.policy_flops = NativeMLP.flops(self.rows, self.num_obs, UNITS, self.num_act)
        # the 27-dof step writes the policy's first-layer input itself (ppenv_ta_sim_set_policy_input: normalised, clamped fp16 rows next to
        # obs_buf), so the rollout step has no normalise-and-pad launch; the 7-dof envs have no such output and keep the launch
        self.prepared = hasattr(self.env, "set_policy_input")
        if self.prepared:
            try:
                self.net.attach_env(self.env)
            except Exception:          # a 27-dof kernel other than the chain-wave one (PPENV_TA_KERNEL): it has no such output, the launch stays
                self.prepared = False

    def forward(self, counter=None):
        if counter is None:
            return self.net.forward(self.obs_buf, prepared=self.prepared)
        return self.net.forward(self.obs_buf, prepared=self.prepared,
                                sample=dict(actions=self.action_buf, sigma=self.sigma, seed=0, counter=counter, neglogp=self.neglogp))

    def step(self, s, slot=None, t=None):
        kw = {}
        if self.into is not None and slot is not None:
            kw = dict(rew=self.into["rew"][slot, t], reset=self.into["done"][slot, t])
            if self.into["obs"] is not None:
                kw["obs"] = self.into["obs"][slot, t]
        if self.rollout:
            # normalise obs -> actor + critic forward -> heads + Normal(mu, sigma) draw + clamp + neglogp -> env.step on the drawn actions
            # (the draw's counter is baked into a captured graph: a replay repeats its 32 counters — a throughput run, not training)
            self.forward(counter=s + 1)
            if "obs" in kw:       # the policy reads the env's own obs_buf; a gathered copy of the row goes to the slice
                obs_slice = kw.pop("obs")
                self.env.step(self.action_buf, **kw)
                obs_slice.copy_(self.obs_buf)
            else:
                self.env.step(self.action_buf, **kw)
        else:
            self.env.step(self.pool[s & 7], **kw)

    def capture(self, lengths, slots=(None,)):
        """One HIP graph per (launch-sequence length, output slot).  Captured BEFORE any process group exists: a capture fails if
        another thread of the process (the RCCL watchdog) touches the runtime while it is open."""
        torch = self.torch
        with torch.no_grad():
            for s in range(HORIZON):       # lazy initialisation (streams, workspaces) must not happen inside the capture
                self.step(s, slots[0], s)
            torch.cuda.synchronize(self.device)
            for length in lengths:
                for slot in slots:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        for s in range(length):    # HORIZON is a multiple of the pool size: every replayed sequence is the eager one
                            self.step(s, slot, s)
                    self.graphs[(length, slot)] = g
            torch.cuda.synchronize(self.device)

    def kernel_region_us(self, warm, launches, regions, fn=None):
        """Average duration of one launch sequence element: regions of back-to-back launches with nothing else on the stream,
        bracketed by HIP events on the stream the kernels are launched on (torch's current stream); the median region."""
        torch = self.torch
        g = self.graphs.get((HORIZON, None)) or self.graphs.get((HORIZON, 0))

        def region(k):
            if fn is not None:
                for _ in range(k):
                    fn()
            elif g is not None:
                for _ in range(k // HORIZON):
                    g.replay()
            else:
                for s in range(k):
                    self.step(s)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.no_grad():
            region(warm)
            torch.cuda.synchronize(self.device)
            out = []
            for _ in range(regions):
                ev0.record()
                region(launches)
                ev1.record()
                torch.cuda.synchronize(self.device)
                out.append(ev0.elapsed_time(ev1) * 1e3 / launches)
        return sorted(out)[len(out) // 2], out

    def close(self):
        self.torch.cuda.synchronize(self.device)
        self.graphs.clear()
        self.env.close()


def hbm_roofline(variant, n, kernel_us, region_us=None, region=None, observed_kernel=None):
    """observed_kernel: what the handle says it launches (ppenv_step_kernel_name / ppenv_ta_sim_kernel_name)."""
    algo = ALGO_BYTES[variant]
    achieved = algo * n / (kernel_us * 1e-6) / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": pmc_traffic(n, variant, observed_kernel), "kernel": observed_kernel or kernel_name(variant), "avg_kernel_us": kernel_us,
         "algorithmic_bytes_per_launch": algo * n,
         "traffic_source": "committed rocprofv3 --pmc passes under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE), not a counter read in this run"}
    if region_us is not None:
        r["region_us"], r["region"] = region_us, region
    sec = valu_roofline(variant, n, kernel_us)
    if sec is not None:
        r["secondary"] = sec        # the binding ceiling (VALU issue) beside the formal one (HBM): DESIGN.md §6
    return r


def cpu_baseline_rollout(num_envs, cores, seconds=4.0):
    """CPU row of the rollout config: the oracle's 27-dof step + the same MLP pair as a plain PyTorch fp32 forward on the host cores."""
    import numpy as np
    import torch
    from isaacgym_amd import scene
    from oracle import binding as ob
    ob.build()
    n = num_envs
    torch.set_num_threads(cores)
    cfg, model, p = scene.build_ta_scene(n), scene.build_ta_model(), scene.build_ta_params(n)
    root = np.zeros((n, 3, 13), np.float32)
    for a in range(3):
        root[:, a, :7] = np.array(list(p.init_root[a]))
    root[:, 2, 7:10] = (-5.0, 0.0, 1.5)
    dof = np.zeros((n, 27, 2), np.float32)
    irb = ob.ta_forward_kinematics(model, root, dof)
    flags, episode, progress = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64)

    def mlp(n_out):
        layers, d = [], 313
        for u in UNITS:
            layers += [torch.nn.Linear(d, u), torch.nn.ELU()]
            d = u
        layers.append(torch.nn.Linear(d, n_out))
        return torch.nn.Sequential(*layers)
    torch.manual_seed(0)
    actor, critic = mlp(27), mlp(1)
    obs = torch.zeros(n, 313)
    t0, steps = time.perf_counter(), 0
    with torch.no_grad():
        while True:
            x = torch.clamp(obs, -5.0, 5.0)
            mu = actor(x)
            critic(x)
            act = torch.clamp(mu + torch.randn_like(mu), -1.0, 1.0).numpy()
            rb, frc, pvx = ob.ta_simulate(cfg, model, act, root, dof, threads=cores)
            o, _, _ = ob.ta_post_physics_step(p, rb, irb, root, dof, frc, pvx, None, flags, episode, progress)
            obs = torch.from_numpy(o)
            steps += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or steps >= 200:
                break
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} rollout steps at num_envs={n}: PyTorch fp32 MLP pair on {cores} threads + the oracle's 27-dof step, {dt:.1f} s"}


def learner_config(device, m, cores, cpu=True):
    """The learner's side of BASELINE configs[4] (train.py + RL-Games PPO on the 27-dof task): one minibatch of the a2c network —
    RunningMeanStd update + forward + backward of the actor / critic pair — on the native kernels (isaacgym_amd.policy.NativeMLPLearner).
    m: minibatch rows.  The reference's yaml carries a debug value (`minibatch_size: 4 # 8192`, cfg/train/HumanoidPingpongTiltG1PPO.yaml:74);
    rows here: its commented production value 8192 and the 32768 of a 4096-env x 32-step horizon split in four."""
    import torch
    from isaacgym_amd.policy import NativeMLP, NativeMLPLearner, RunningMeanStd
    num_obs, num_act = 313, 27

    def mlp(n_out):
        layers, d = [], num_obs
        for u in UNITS:
            layers.append(torch.nn.Linear(d, u))
            d = u
        layers.append(torch.nn.Linear(d, n_out))
        return [(l.weight, l.bias) for l in layers]
    torch.manual_seed(0)
    learner = NativeMLPLearner(mlp(num_act), mlp(1), num_obs, device)
    rms = RunningMeanStd(num_obs, device)
    learner.attach_running_mean_std(rms)
    gen = torch.Generator(device=device).manual_seed(1)
    obs = torch.randn(m, num_obs, device=device, generator=gen) * 2.0
    d_head = torch.randn(m, num_act + 1, device=device, generator=gen)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        out = []
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ev0.record()
            for _ in range(reps):
                fn()
            ev1.record()
            torch.cuda.synchronize(device)
            out.append(ev0.elapsed_time(ev1) * 1e3 / reps)
        return sorted(out)[1]
    with torch.no_grad():
        fwd_us = timed(lambda: learner.forward(obs, update_stats=True))
        bwd_us = timed(lambda: learner.backward(d_head))
        sync_us = timed(learner.sync_weights, reps=5)
    dims = [num_obs] + UNITS
    fwd_fl = NativeMLP.flops(m, num_obs, UNITS, num_act)
    bwd_fl = 2 * fwd_fl - 2 * m * 2 * dims[0] * dims[1]                    # dW for every layer, dX for every layer but the first (observations need no gradient)
    tf = (fwd_fl + bwd_fl) / ((fwd_us + bwd_us) * 1e-6) / 1e12
    row = {"name": f"c5_TA_learner_minibatch_{m}", "baseline_config": "BASELINE.json configs[4]", "variant": "TA", "minibatch_rows": m,
           "workload": "a2c network of cfg/train/HumanoidPingpongTiltG1PPO.yaml:10-31 (actor + critic [2048, 1536, 1024, 1024, 512, 512], ELU, 313 obs, 27 actions): "
                       "RunningMeanStd update + forward + backward of one minibatch; random-init weights, synthetic rows",
           "dtype": "f16 operands / f32 accumulation, f32 weight and bias gradients",
           "us_forward_incl_stats_update": fwd_us, "us_backward": bwd_us, "us_weight_cast_per_optimizer_step": sync_us,
           "rows_per_s_forward_backward": m / ((fwd_us + bwd_us) * 1e-6), "gflop_forward": fwd_fl / 1e9, "gflop_backward": bwd_fl / 1e9,
           "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS, "traffic": None,
                        "kernel": "mlp_layer_pp_kernel (forward, dX) + mlp_dw_kernel (dW): forward + backward launch sequence",
                        "avg_kernel_us": fwd_us + bwd_us, "flops_per_launch_sequence": fwd_fl + bwd_fl}}
    del learner
    torch.cuda.empty_cache()
    if cpu:
        try:
            torch.set_num_threads(cores)
            mc = 512
            torch.manual_seed(0)

            def net(n_out):
                layers, d = [], num_obs
                for u in UNITS:
                    layers += [torch.nn.Linear(d, u), torch.nn.ELU()]
                    d = u
                layers.append(torch.nn.Linear(d, n_out))
                return torch.nn.Sequential(*layers)
            actor, critic = net(num_act), net(1)
            x, g = torch.randn(mc, num_obs), torch.randn(mc, num_act + 1)
            t0, it = time.perf_counter(), 0
            while True:
                out = torch.cat([actor(x), critic(x)], dim=1)
                out.backward(g)
                it += 1
                dt = time.perf_counter() - t0
                if dt >= 4.0 or it >= 50:
                    break
            row["cpu_baseline"] = {"value": mc * it / dt, "unit": "minibatch rows/s (forward + backward)", "cores": cores, "kind": "port",
                                   "sample": f"{it} forward + backward passes of {mc} rows, PyTorch fp32 autograd on {cores} threads, {dt:.1f} s"}
        except Exception as e:   # noqa: BLE001
            row["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    return row


def secondary_configs(device, only=None, cpu=True):
    """The other single-GPU BASELINE.json configs, a few seconds each, for the same JSON line (`configs`): configs[1] 3-actor at
    N = 4096 (T3 and TT semantics), configs[3] 4-actor tilt at its 8192 envs per GPU, configs[4] 3-actor all-dof at its 4096 envs
    per GPU — the step alone and the rollout step incl. the native policy forward.  Each row: avg_kernel_us from event-bracketed
    regions of back-to-back launches (3 x 320 after 128), roofline (HBM bytes for the steps, dense-fp16 MFMA flops for the
    forward), and a bounded cpu_baseline row."""
    import torch
    rows = []
    cores = usable_cores()
    plan = [("c2_T3_4096", 2, "T3", 4096, False), ("c2_TT_4096", 2, "TT", 4096, False), ("c4_T4_8192", 4, "T4", 8192, False),
            ("c5_TA_4096_step", 5, "TA", 4096, False), ("c5_TA_4096_rollout", 5, "TA", 4096, True)]
    for name, cfg_no, variant, n, rollout in plan:
        if only and name not in only:
            continue
        try:
            rows.append(_secondary_row(device, name, cfg_no, variant, n, rollout, rows, cores, cpu))
        except Exception as e:   # noqa: BLE001 - a secondary row must never cost the headline line
            rows.append({"name": name, "error": f"{type(e).__name__}: {e}"})
            torch.cuda.empty_cache()
    for m in (8192, 32768):
        if only and f"c5_TA_learner_minibatch_{m}" not in only:
            continue
        try:
            rows.append(learner_config(device, m, cores, cpu=cpu))
        except Exception as e:   # noqa: BLE001
            rows.append({"name": f"c5_TA_learner_minibatch_{m}", "error": f"{type(e).__name__}: {e}"})
            torch.cuda.empty_cache()
    return rows


def run_secondary_child(args, timeout_s=420):
    """secondary_configs() in a child process: `python bench.py --only-config ALL` -> its `configs` list (or one error row)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--only-config", "ALL"] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ))
        try:
            stdout, _ = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            proc.kill()                      # exactly the process started above
            proc.communicate()
            return [{"error": f"secondary configs: the child process exceeded {timeout_s} s and was killed"}]
        for line in reversed(stdout.decode(errors="replace").splitlines()):
            if line.startswith("{"):
                return json.loads(line)["configs"]
        return [{"error": f"secondary configs: the child process ended with code {proc.returncode} and no JSON line"}]
    except Exception as e:   # noqa: BLE001 - the headline line is printed whatever happens to the secondary rows
        return [{"error": f"{type(e).__name__}: {e}"}]


def configs_summary(rows):
    """name -> the few scalars of each secondary row (flat and short: what the driver's record and a reader's eye keep)."""
    out = {}
    for r in rows:
        name = r.get("name", "error")
        if "error" in r:
            out[name] = {"error": r["error"][:120]}
            continue
        roof = r.get("roofline", {})
        e = {"frac": round(roof.get("frac", 0.0), 4), "bound": roof.get("bound")}
        for k_out, k_in in (("env_steps_per_s", "env_steps_per_s"), ("avg_kernel_us", "avg_kernel_us"), ("us_per_rollout_step", "us_per_rollout_step"),
                            ("us_policy_forward", "us_policy_forward"), ("us_forward", "us_forward_incl_stats_update"), ("us_backward", "us_backward"),
                            ("rows_per_s", "rows_per_s_forward_backward")):
            if r.get(k_in) is not None:
                e[k_out] = round(r[k_in], 3 if "us" in k_out else 0)
        sec = roof.get("secondary")
        if sec:
            e["valu_issue_frac"], e["wait_frac"] = round(sec["frac"], 4), round(sec.get("wait_frac", 0.0), 3)
        if roof.get("traffic"):
            e["traffic_over_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes_per_launch"], 3)
        cb = r.get("cpu_baseline", {})
        if "value" in cb:
            e["cpu_baseline"] = round(cb["value"], 0)
        out[name] = e
    return out


def _secondary_row(device, name, cfg_no, variant, n, rollout, rows, cores, cpu):
    """One row of secondary_configs: an env-step config (HBM roofline) or the rollout config (MFMA roofline of its policy forward)."""
    import torch
    w = Workload(variant, n, device, rollout=rollout)
    w.capture([HORIZON])
    us, regions = w.kernel_region_us(128, 320, 3)
    row = {"name": name, "baseline_config": f"BASELINE.json configs[{cfg_no - 1}]", "variant": variant, "num_envs": n,
           "workload": WORKLOAD_NAMES[variant] + (", rollout step = normalise + actor/critic forward + action draw + env step" if rollout else ""),
           "launch": f"HIP graph of {HORIZON} steps, replayed", "dtype": "f32" if not rollout else "f32 env step, f16 operands / f32 accumulation in the policy",
           "region": "3 x 320 launches after 128 (median)"}
    if not rollout:
        row.update({"avg_kernel_us": us, "env_steps_per_s": n / (us * 1e-6), "agent_steps_per_s": w.rows / (us * 1e-6),
                    "roofline": hbm_roofline(variant, n, us, regions, observed_kernel=w.kernel)})
    else:
        fwd_us, _ = w.kernel_region_us(16, 64, 3, fn=w.forward)            # the eight launches of one forward, eager, back to back
        tf = w.policy_flops / (fwd_us * 1e-6) / 1e12
        step_us = rows[-1].get("avg_kernel_us") if rows and rows[-1]["name"] == "c5_TA_4096_step" else None
        row.update({"us_per_rollout_step": us, "env_steps_per_s": n / (us * 1e-6), "us_policy_forward": fwd_us,
                    "us_env_step_alone": step_us, "policy_gflop_per_step": w.policy_flops / 1e9,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
                                 "traffic": None, "kernel": "mlp_layer_pp_kernel / mlp_layer_pp1_kernel (policy forward: 8 launches)",
                                 "avg_kernel_us": fwd_us, "flops_per_launch_sequence": w.policy_flops}})
    w.close()
    del w
    torch.cuda.empty_cache()
    if cpu:
        try:
            row["cpu_baseline"] = cpu_baseline_rollout(n, cores, seconds=6.0) if rollout else cpu_baseline(n, target_seconds=2.5, variant=variant)
        except Exception as e:   # noqa: BLE001 - a CPU row must not cost the GPU rows
            row["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--num-envs", type=int, default=None, help="envs per GPU (default 16384; 4096 for --workload rollout)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", default=None, choices=["TT", "TN", "T3", "T4", "TA"],
                    help="task variant; the headline workload is TT (BASELINE.json configs[2]), the others are parity-test cases")
    ap.add_argument("--workload", default="step", choices=["step", "rollout"],
                    help="step (default): the fused env step on synthetic actions, the BASELINE metric.  rollout: BASELINE configs[4]'s per-GPU "
                         "slice — the 27-dof env step driven by the native policy forward (4096 envs per GPU), weak-scaled by --gpus N like the step")
    ap.add_argument("--gather", default="none", choices=["none", "rew", "obs"],
                    help="the north star's episodic RCCL gather for a central learner (isaacgym_amd.distributed.RolloutGather): rew = rewards + dones of "
                         "every horizon, once per horizon; obs = also every step's observation rows, once per step (eager launches).  Asynchronous, two "
                         "rotating slots.  none (default): data-parallel learners, the reference's own mode — no data-path collective")
    ap.add_argument("--gather-pad", type=int, default=0, help="pad every shard to at least this many rows (exercises the ragged path with one rank)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph of one horizon")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary BASELINE configs (`configs` in the JSON line); they run at N=1 only")
    ap.add_argument("--only-config", action="append", default=None,
                    help="run only this secondary config (name as in `configs`; repeatable) and print its row(s) alone — for profiling one at a time")
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU rehearsal of the rank plumbing only (launch, rendezvous, barrier, max-over-ranks, rank-0 line): no env is "
                         "created and nothing is measured; used by tests/test_bench_launch.py")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the fixed pre-warm launches before the W warm-up steps")
    ap.add_argument("--dist-backend", default="nccl", help="nccl = RCCL (default); gloo only to rehearse the rank logic on one GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the process group and issue the per-horizon all-reduces (and gathers) even with ONE rank: the RCCL code path "
                         "(communicator init with device_id, asynchronous collectives beside the graph replays, barrier, max-reduce) on a one-GPU box")
    args = ap.parse_args()
    rollout = args.workload == "rollout"
    if args.variant is None:
        args.variant = "TA" if rollout else VARIANT
    if args.num_envs is None:
        args.num_envs = 4096 if rollout else NUM_ENVS
    if rollout and args.variant == "T4":
        sys.exit("--workload rollout drives one policy row per env: variants TA (BASELINE configs[4]) or TT / TN / T3")

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # `python bench.py --gpus N` as invoked: this parent touches neither torch nor the GPU; it starts one fresh child process
        # per rank with the launcher's environment (what `python -m torch.distributed.run --nproc-per-node N` would set), relays
        # rank 0's JSON line and exits with the worst child's return code.  (A process that has initialised the GPU must never be
        # replaced by another program; children are started, not exec'ed.)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    if args.rehearse:
        sys.exit(rehearse(args))

    import torch
    from isaacgym_amd import _lib
    from isaacgym_amd import distributed as D

    if not os.path.exists(_lib.LIB_PATH):      # a checkout without the built library (it is git-ignored): local rank 0 compiles it, the others wait
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            _lib.build()
        else:
            t_end = time.time() + 900
            while not os.path.exists(_lib.LIB_PATH) and time.time() < t_end:
                time.sleep(1.0)
    _lib.lib()   # fail loudly when the HIP extension is missing (there is no CPU path)
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                          # under a launcher the world size is authoritative
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.dist_backend == "nccl":
        sys.exit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible; one process per GPU is required")
    device = torch.device("cuda", local_rank % max(ndev, 1))   # the modulo only matters for the gloo rehearsal
    torch.cuda.set_device(device)

    if args.only_config:
        rows = secondary_configs(device, only=None if args.only_config == ["ALL"] else set(args.only_config), cpu=not args.no_cpu_baseline)
        print(json.dumps({"configs": rows}), flush=True)
        return

    n = args.num_envs
    gather = args.gather != "none"
    per_step_gather = args.gather == "obs"
    use_graph = not args.no_graph and not per_step_gather     # a per-step collective sits between the launches: eager
    GSLOTS = 2
    w = Workload(args.variant, n, device, rank=rank, world=world, rollout=rollout, into_depth=GSLOTS if gather else 0, into_obs=per_step_gather)
    env = w.env
    if use_graph:
        # one graph per launch-sequence length in use: the horizon, and what --warmup / --steps leave over after whole horizons
        # (a driver that asks for fewer steps than a horizon still gets device-paced launches); per output slot in the gather modes
        w.capture(sorted({HORIZON, args.warmup % HORIZON, args.steps % HORIZON} - {0}), slots=tuple(range(GSLOTS)) if gather else (None,))

    if dist is not None:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)
    # what the reference prints every 40 steps (TT:763-766) + finished episodes; all-reduced over the ranks
    stats = D.AsyncHorizonStats(env, force=args.force_dist) if args.variant != "TA" else None
    rg = D.RolloutGather(w.rows, device, depth=GSLOTS, force=args.force_dist, pad_to=args.gather_pad) if gather else None
    hz = [0]        # horizons issued: picks the output slot

    def slot_now():
        return (hz[0] % GSLOTS) if gather else None

    def horizon_done():
        if stats is not None:
            stats.push()               # one reduction launch + one asynchronous 4-double all-reduce over RCCL
        if rg is not None:             # rewards + dones of the finished horizon: two asynchronous all-gathers into slot hz % 2
            sl = slot_now()
            rg.push(sl, [w.into["rew"][sl], w.into["done"][sl]], env_dims=[1, 1])
            hz[0] += 1
            rg.wait(slot_now())        # the slot the next horizon writes into: its previous gathers must have read it

    obs_rg = D.RolloutGather(w.rows, device, depth=4, force=args.force_dist, pad_to=args.gather_pad) if per_step_gather else None
    obs_calls = [0]

    def eager_step(s):
        sl, t = slot_now(), s % HORIZON
        w.step(s, sl, t)
        if obs_rg is not None:         # every step's observation rows: one asynchronous all-gather per step
            obs_rg.push(obs_calls[0] % 4, [w.into["obs"][sl, t]])
            obs_calls[0] += 1

    @torch.no_grad()
    def run(k):
        done = 0
        if use_graph:
            for _ in range(k // HORIZON):
                w.graphs[(HORIZON, slot_now())].replay()
                horizon_done()
            done = (k // HORIZON) * HORIZON
            if (k - done, slot_now()) in w.graphs:
                w.graphs[(k - done, slot_now())].replay()
                done = k
        for s in range(done, k):
            eager_step(s)
            if (s + 1) % HORIZON == 0:
                horizon_done()

    if not args.no_prewarm:
        # Setup, like the graph capture above: PREWARM launches so that a short run (the driver's --steps 20 --warmup 5 is one
        # 0.3 ms graph replay) is not a measurement of the clock ramp.  Not counted in `warmup`; the W warm-up steps follow.
        for _ in range(PREWARM // HORIZON):
            if use_graph:
                w.graphs[(HORIZON, slot_now())].replay()
            else:
                run(HORIZON)
        torch.cuda.synchronize(device)
    run(args.warmup)
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps)
    ev1.record()
    while not ev1.query():     # poll first: a blocking wait's wake-up costs tens of microseconds, a fifth of the driver's 20-step (0.23 ms) region
        pass
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)

    if dist is not None:
        t = torch.tensor([wall], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    # roofline: average duration of the step kernel alone — regions of back-to-back launches with nothing else on the stream,
    # bracketed by HIP events on the stream the kernel is launched on (torch's current stream).  The region does not depend on
    # --steps / --warmup: ROOFLINE_WARM launches to settle the clocks, then ROOFLINE_REGIONS regions of ROOFLINE_LAUNCHES
    # launches each; the median region is reported (profiles/*_kernel_stats.csv is the same command under rocprofv3).
    if rg is not None:
        for sl in range(GSLOTS):
            rg.wait(sl)
    if obs_rg is not None:
        for sl in range(4):
            obs_rg.wait(sl)
    kernel_us, region_us = w.kernel_region_us(ROOFLINE_WARM, ROOFLINE_LAUNCHES, ROOFLINE_REGIONS)
    fwd_us = None
    if rollout:
        fwd_us, _ = w.kernel_region_us(32, 96, 5, fn=w.forward)
    gather_info = None
    if gather:
        # the collectives' own cost, blocking, on an otherwise idle device: issue + completion of one horizon's (one step's) gathers
        def one_call(g_, tensors, dims):
            g_.push(0, tensors, env_dims=dims)
            g_.wait(0)
            torch.cuda.synchronize(device)
        per = {}
        for label, g_, tensors, dims in (("horizon_rew_done", rg, [w.into["rew"][0], w.into["done"][0]], [1, 1]),) + \
                ((("step_obs", obs_rg, [w.into["obs"][0, 0]], [0]),) if per_step_gather else ()):
            one_call(g_, tensors, dims)
            tc = time.perf_counter()
            for _ in range(20):
                one_call(g_, tensors, dims)
            per[label] = {"us_per_call_blocking": (time.perf_counter() - tc) / 20 * 1e6,
                          "bytes_sent_per_rank_per_call": sum(x.numel() * x.element_size() for x in tensors),
                          "bytes_received_per_rank_per_call": sum(x.numel() * x.element_size() for x in tensors) * rg.world}
        bytes_per_step = per["horizon_rew_done"]["bytes_sent_per_rank_per_call"] / HORIZON + (per["step_obs"]["bytes_sent_per_rank_per_call"] if per_step_gather else 0)
        gather_info = {"mode": args.gather, "backend": args.dist_backend if dist is not None else "none (one rank, local copy)", "ranks": rg.world,
                       "shard_rows": rg.counts, "padded_rows": rg.m, "ragged": rg.ragged, "bytes_per_step_per_rank": bytes_per_step,
                       "horizon_gathers_issued": hz[0], "step_gathers_issued": obs_calls[0], "collectives": per,
                       "what": "asynchronous all_gather_into_tensor per tensor into rank-major buffers, two rotating slots; rewards + dones "
                               "[32, rows] once per horizon" + ("; observation rows [rows, num_obs] once per step" if per_step_gather else "")}
    if stats is not None:
        stats.push()
        final_stats = stats.latest().cpu().tolist()
    else:
        final_stats = [float(env.rew_buf.mean()), float(env.progress_buf.float().mean()), float(env.state.episode.sum())]

    if rank == 0:
        total_env_steps = n * world * args.steps
        if not rollout:
            roof = hbm_roofline(args.variant, n, kernel_us, region_us, f"{ROOFLINE_REGIONS} x {ROOFLINE_LAUNCHES} launches after {ROOFLINE_WARM} (median)",
                                observed_kernel=w.kernel)
            roof["timed_region_us_per_step"] = dev_ms * 1e3 / args.steps
            metric = "env-steps/sec at N_envs=16384 (1/2/4/8 GPUs) + achieved HBM GB/s vs roofline"
            what = "random U(-1,1) actions, 2 physics substeps per step, fused step kernel"
        else:
            tf = w.policy_flops / (fwd_us * 1e-6) / 1e12
            roof = {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS, "traffic": None,
                    "kernel": "mlp_layer_pp_kernel / mlp_layer_pp1_kernel (policy forward: 8 launches)", "avg_kernel_us": fwd_us,
                    "flops_per_launch_sequence": w.policy_flops, "us_per_rollout_step_region": kernel_us, "region_us": region_us,
                    "timed_region_us_per_step": dev_ms * 1e3 / args.steps}
            metric = "env-steps/sec incl. policy forward (BASELINE.json configs[4] per-GPU slice: 3-actor all-dof + native policy forward)"
            what = "rollout step = actor/critic MLP forward (MFMA; its normalised fp16 input written by the env step itself) + Normal(mu, sigma) draw + fused env step"
        out = {
            "metric": metric,
            "value": total_env_steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"humanoid_pingpong {WORKLOAD_NAMES[args.variant]}, num_envs={n} per GPU, {what}",
                       "variant": args.variant, "kind": args.workload,
                       "num_envs_per_gpu": n, "global_envs": n * world, "horizon_stats_every": HORIZON,
                       "prewarm_launches": 0 if args.no_prewarm else PREWARM,
                       "launch": "eager" if not use_graph else f"HIP graph of {HORIZON} steps, replayed",
                       "parallelism": f"env-shard x{world}, " + ("no data-path collective" if not gather else f"RCCL all-gather ({args.gather}) for a central learner")},
            "roofline": roof,
            "episode_stats": {"mean_reward_last_step": final_stats[0], "mean_progress": final_stats[1],
                              "episodes_finished": final_stats[2]},
        }
        if gather_info is not None:
            out["gather"] = gather_info
    w.close()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_rollout(n, usable_cores(), seconds=10.0) if rollout else cpu_baseline(n, variant=args.variant)
            if args.variant == VARIANT and not rollout:
                out["cpu_baseline_rows"] = cpu_baseline_rows(out["cpu_baseline"]["cores"])
        if world == 1 and not args.no_configs and not rollout and args.variant == VARIANT and not gather and dist is None:
            # The secondary rows run in a freshly started CHILD process (this script with --only-config), after the headline has been
            # measured and its env closed: a device fault, abort or hang in one of them (each touches other kernels) can then cost
            # only that child — it is killed by exact PID at its time limit — never the headline line printed below.
            out["configs"] = run_secondary_child(args)
            summary = configs_summary(out["configs"])
            out["config"]["configs_summary"] = summary          # inside `config`: the driver's record keeps that object's values
        out["config"]["episode_stats"] = out["episode_stats"]
        if "secondary" in out["roofline"]:
            out["roofline_secondary"] = out["roofline"]["secondary"]
        if "configs" in out:
            out["configs_summary"] = out["config"]["configs_summary"]     # ... and LAST on the line, so a tail of the output ends with it
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
