#!/usr/bin/env python3
"""Soak of the chain-wave 27-dof step (ppenv_ta_step: rigid-body step + reward / reset / 313 observations in one launch) against the
CPU oracle's ta_simulate + ta_post_physics_step, restarted from the oracle's tensors every step — the comparison of
tests/test_ta_physics.py::test_ta_chain_kernel_step_matches_oracle on more envs, steps and seeds, with the envs whose ORACLE result
moves under a 2e-6 jitter of its own inputs set aside (tools/gpu_soak_ta.py).
Run on the GPU box: python tools/gpu_soak_ta_chain.py [n] [steps] [seeds...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PPENV_TA_KERNEL"] = "chain"
import torch  # noqa: E402
import oracle.binding as ob  # noqa: E402
from helpers import assert_close  # noqa: E402
from isaacgym_amd import scene  # noqa: E402
from isaacgym_amd.tensor_api import TAEnv  # noqa: E402
from test_ta_physics import TOL, _ta_obs_atol, check_step  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seeds = [int(a) for a in sys.argv[3:]] or [31, 32]
ob.build()
cfg, m = scene.build_ta_scene(n), scene.build_ta_model()
oa = np.delete(_ta_obs_atol(), 120)
tot = bad = excluded = resets = 0


def sensitive(act, root0, dof0, root1, dof1, rng, rel=2e-6):
    out = np.zeros(n, bool)
    for _ in range(2):
        rj = (root0 * (1 + rel * rng.uniform(-1, 1, root0.shape))).astype(np.float32)
        dj = (dof0 * (1 + rel * rng.uniform(-1, 1, dof0.shape))).astype(np.float32)
        ob.ta_simulate(cfg, m, act, rj, dj, threads=16)
        out |= (np.abs(dj[..., 1] - dof1[..., 1]) > 0.3 * TOL["qd"]).any(axis=1)
        out |= (np.abs(dj[..., 0] - dof1[..., 0]) > 0.3 * TOL["q"]).any(axis=1)
        out |= (np.abs(rj[:, 0, 7:13] - root1[:, 0, 7:13]) > 0.3 * TOL["root_vel"]).any(axis=1)
    return out


for seed in seeds:
    env = TAEnv(n, device="cuda:0", seed=seed, env={"episodeLength": 60}, materialize_rb=True)
    assert env.sim.kernel == "chain"
    p = env.params
    root, dof = env.root_states.cpu().numpy().copy(), env.dof_states.cpu().numpy().copy()
    irb = env.initial_rb_states.cpu().numpy().copy()
    flags, episode, progress = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64)
    rng = np.random.default_rng(seed + 100)
    for t in range(steps):
        if t % 4 == 0:
            act = rng.uniform(-1.2, 1.2, (n, 27)).astype(np.float32)
            act[: n // 3] *= 0.1
        env.root_states.copy_(torch.from_numpy(root)); env.dof_states.copy_(torch.from_numpy(dof))
        env.state.flags.copy_(torch.from_numpy(flags.view(np.int32))); env.state.episode.copy_(torch.from_numpy(episode.view(np.int32)))
        env.state.progress_buf.copy_(torch.from_numpy(progress))
        env.step(torch.from_numpy(act).cuda())
        root0, dof0 = root.copy(), dof.copy()
        rb, frc, pvx = ob.ta_simulate(cfg, m, act, root, dof, threads=16)
        root1, dof1 = root.copy(), dof.copy()
        obs, rew, reset = ob.ta_post_physics_step(p, rb, irb, root, dof, frc, pvx, None, flags, episode, progress)
        g_rb = env._rb_states.cpu().numpy()
        keep = ~(np.abs(g_rb[:, 41, 7:10] - rb[:, 41, 7:10]).max(axis=1) > 1e-3)
        keep &= ~sensitive(act, root0, dof0, root1, dof1, rng)
        excluded += int((~keep).sum())
        try:
            g_reset, g_prog, g_flags = env.reset_buf.cpu().numpy(), env.progress_buf.cpu().numpy(), env.state.flags.cpu().numpy().view(np.uint32)
            for what, gv, ov in (("reset_buf", g_reset, reset), ("progress_buf", g_prog, progress), ("flags", g_flags, flags)):
                diff = np.nonzero(gv != ov)[0]
                # a discrete decision may legitimately differ in an env set aside as sensitive (its continuous state sits on a threshold)
                assert not keep[diff].any(), f"seed {seed} step {t}: {what} differs in retained envs {diff[keep[diff]][:8]} (and {int((~keep[diff]).sum())} set-aside ones)"
                if diff.size:
                    print(f"seed {seed} step {t}: {what} differs in {diff.size} set-aside env(s) {diff[:4]} — not counted", flush=True)
            check_step((env.root_states.cpu().numpy()[keep], env.dof_states.cpu().numpy()[keep], g_rb[keep][:, :40], env.dof_force_tensor.cpu().numpy()[keep]),
                       (root[keep], dof[keep], rb[keep][:, :40], frc[keep]), f"seed {seed} step {t}")
            assert_close(np.delete(env.obs_buf.cpu().numpy(), 120, axis=1)[keep], np.delete(obs, 120, axis=1)[keep], f"seed {seed} step {t}: obs", atol=oa)
            assert_close(env.rew_buf.cpu().numpy()[keep], rew[keep], f"seed {seed} step {t}: rew", atol=1e-4 * 3000.0 * 0.5)
        except AssertionError as e:
            bad += 1
            print(str(e)[:300], flush=True)
        flags[~keep] = env.state.flags.cpu().numpy().view(np.uint32)[~keep]
        tot += int(keep.sum())
        resets += int(reset.sum())
    assert env.sim.status == 0
    env.close()
    print("seed", seed, "done", flush=True)
print("chain-wave 27-dof step: env-steps compared", tot, "excluded (discrete ball contact, or the oracle itself moves under a 2e-6 jitter)", excluded,
      "resets", resets, "steps with a violation", bad)
