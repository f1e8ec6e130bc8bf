"""The native rollout collector (isaacgym_amd/collector.py): horizon-major buffers written in place by the env step, the heads launch
and ppenv_gae — against the same loop done by hand and GAE in plain torch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _net(torch, num_obs, num_act, device):
    from isaacgym_amd.policy import NativeMLP
    gen = torch.Generator().manual_seed(0)

    def mlp(n_out):
        d, out = num_obs, []
        for u in (256, 128, 64) + (n_out,):
            out.append(((torch.rand(u, d, generator=gen) * 2 - 1) / np.sqrt(d), torch.zeros(u)))
            d = u
        return out
    return NativeMLP(mlp(num_act), mlp(1), num_obs, device, mean=torch.zeros(num_obs), var=torch.ones(num_obs))


def test_gae_matches_the_recurrence_in_torch():
    import torch
    from isaacgym_amd.collector import gae
    h, n = 32, 1000
    gen = torch.Generator(device="cuda").manual_seed(1)
    rew = torch.randn(h, n, device="cuda", generator=gen) * 50
    head = torch.randn(h + 1, n, 28, device="cuda", generator=gen)       # values as a strided column view, as the collector holds them
    val = head[:, :, 27]
    done = (torch.rand(h, n, device="cuda", generator=gen) < 0.05).to(torch.int64)
    adv, ret = gae(rew, val, done, 0.99, 0.95, 0.01)
    want = torch.zeros(h, n, device="cuda", dtype=torch.float64)
    run = torch.zeros(n, device="cuda", dtype=torch.float64)
    for t in reversed(range(h)):
        nd = 1.0 - done[t].double()
        delta = 0.01 * rew[t].double() + 0.99 * val[t + 1].double() * nd - val[t].double()
        run = delta + 0.99 * 0.95 * nd * run
        want[t] = run
    assert torch.allclose(adv.double(), want, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ret.double(), want + val[:h].double(), rtol=1e-5, atol=1e-5)


def test_collector_equals_the_loop_done_by_hand():
    """Two horizons: every buffer equals what a second env + the same network produce when stepped one call at a time with copies."""
    import torch
    from isaacgym_amd.collector import RolloutCollector, gae
    from isaacgym_amd.policy import sample_actions
    from isaacgym_amd.tensor_api import TAEnv
    n, h = 512, 8
    env, ref = TAEnv(n, device="cuda:0", seed=5, env={"episodeLength": 12}), TAEnv(n, device="cuda:0", seed=5, env={"episodeLength": 12})
    net = _net(torch, 313, 27, "cuda:0")
    sigma = torch.full((27,), 0.4, device="cuda")
    col = RolloutCollector(env, net, horizon=h, sigma=sigma, seed=3)
    counter = 0
    for horizon in range(2):
        col.collect()
        torch.cuda.synchronize()
        obs, acts, nls, vals, rews, dns = [ref.obs_buf.clone()], [], [], [], [], []
        for t in range(h):
            counter += 1
            mu, v = net.forward(ref.obs_buf)
            a, nl = torch.zeros(n, 27, device="cuda"), torch.zeros(n, device="cuda")
            sample_actions(a, mu, sigma, 3, counter, -1.0, 1.0, nl)
            vals.append(v[:, 0].clone())
            ref.step(a)
            obs.append(ref.obs_buf.clone()); acts.append(a); nls.append(nl); rews.append(ref.rew_buf.clone()); dns.append(ref.reset_buf.clone())
        vals.append(net.forward(ref.obs_buf)[1][:, 0].clone())
        assert torch.equal(col.obs, torch.stack(obs)) and torch.equal(col.actions, torch.stack(acts)) and torch.equal(col.neglogp, torch.stack(nls))
        assert torch.equal(col.values, torch.stack(vals)) and torch.equal(col.rewards, torch.stack(rews)) and torch.equal(col.dones, torch.stack(dns))
        adv, ret = gae(torch.stack(rews), torch.stack(vals), torch.stack(dns), 0.99, 0.95, 0.01)
        assert torch.equal(col.advantages, adv) and torch.equal(col.returns, ret)
        assert int(col.dones.sum()) > 0 or horizon == 0                     # episodes of 12 steps: the second horizon sees resets
        col.next_horizon()
    env.close(); ref.close()


@pytest.mark.parametrize("variant", ["TT", "T4"])
def test_collector_on_the_7dof_tasks_equals_the_loop_done_by_hand(variant):
    """The same for PPEnv (ppenv_step_into redirects obs / rew / reset per call); T4 has two actor rows per env."""
    import torch
    from isaacgym_amd import scene
    from isaacgym_amd.collector import RolloutCollector
    from isaacgym_amd.env import PPEnv
    from isaacgym_amd.policy import sample_actions
    n, h = 700, 6
    mk = lambda: PPEnv(scene.build_config(variant, num_envs=n, seed=4), device="cuda:0")
    env, ref = mk(), mk()
    rows = env.num_rows
    net = _net(torch, 80, 7, "cuda:0")
    sigma = torch.full((7,), 0.5, device="cuda")
    col = RolloutCollector(env, net, horizon=h, sigma=sigma, seed=9)
    counter = 0
    for horizon in range(3):
        col.collect()
        torch.cuda.synchronize()
        obs, acts, rews, dns = [ref.obs_buf.clone()], [], [], []
        for t in range(h):
            counter += 1
            mu, v = net.forward(ref.obs_buf)
            a = torch.zeros(rows, 7, device="cuda")
            sample_actions(a, mu, sigma, 9, counter, -1.0, 1.0)
            ref.step(a)
            obs.append(ref.obs_buf.clone()); acts.append(a); rews.append(ref.rew_buf.clone()); dns.append(ref.reset_buf.clone())
        assert torch.equal(col.obs, torch.stack(obs)) and torch.equal(col.actions, torch.stack(acts))
        assert torch.equal(col.rewards, torch.stack(rews)) and torch.equal(col.dones, torch.stack(dns))
        assert bool(torch.isfinite(col.advantages).all())
        col.next_horizon()
    # the handle's own buffers were not written by the redirected steps, its state was
    assert torch.equal(env.ball, ref.ball) and torch.equal(env.dof_pos, ref.dof_pos) and torch.equal(env.progress_buf, ref.progress_buf)
    env.close(); ref.close()
