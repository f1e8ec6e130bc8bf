"""N2: the policy forward on the MFMA kernel (include/ppenv_policy.h) against a plain PyTorch fp32 MLP on the same weights.
Tolerance: fp16 operands with fp32 accumulation — rtol 1e-2 plus 1e-2 of the tensor's scale (north star for fp16 paths)."""
import numpy as np
import pytest


def _mlp(torch, num_obs, units, n_out, gen):
    layers, d = [], num_obs
    for u in list(units) + [n_out]:
        w = (torch.rand(u, d, generator=gen) * 2 - 1) * (1.0 / np.sqrt(d))       # nn.Linear's default range
        b = (torch.rand(u, generator=gen) * 2 - 1) * (1.0 / np.sqrt(d))
        layers.append((w, b))
        d = u
    return layers


def _ref_forward(torch, layers, x):
    for i, (w, b) in enumerate(layers):
        x = x @ w.t() + b
        if i + 1 < len(layers):
            x = torch.nn.functional.elu(x)
    return x


@pytest.mark.gpu
@pytest.mark.parametrize("m,num_obs,num_act,units", [(4096, 313, 27, (2048, 1536, 1024, 1024, 512, 512)), (1000, 80, 7, (2048, 1536, 1024, 1024, 512, 512)),
                                                      (130, 80, 7, (256, 128)), (77, 33, 5, (100, 50, 36))])   # last: nothing aligned, nothing a tile multiple
@pytest.mark.parametrize("fuse_input", [False, True])
def test_native_mlp_matches_fp32_pytorch(m, num_obs, num_act, units, fuse_input):
    import torch
    from isaacgym_amd.policy import NativeMLP
    gen = torch.Generator().manual_seed(m)
    actor, critic = _mlp(torch, num_obs, units, num_act, gen), _mlp(torch, num_obs, units, 1, gen)
    mean = torch.randn(num_obs, generator=gen) * 0.5
    var = torch.rand(num_obs, generator=gen) * 2 + 0.1
    obs = (torch.randn(m, num_obs, generator=gen) * 2.0 + 0.3)
    obs[:, 5] *= 20.0                                   # a column that hits the +-5 clamp
    net = NativeMLP(actor, critic, num_obs, "cuda:0", mean=mean, var=var, fuse_input=fuse_input)
    mu, value = net.forward(obs.cuda())
    torch.cuda.synchronize()
    x = torch.clamp((obs - mean) / torch.sqrt(var + 1e-5), -5.0, 5.0)
    want_mu, want_v = _ref_forward(torch, actor, x), _ref_forward(torch, critic, x)
    for got, want, what in ((mu.cpu(), want_mu, "mu"), (value.cpu(), want_v, "value")):
        scale = float(want.abs().max())
        err = (got - want).abs()
        tol = 1e-2 * want.abs() + 1e-2 * scale
        assert bool((err <= tol).all()), (what, float(err.max()), scale)
        assert float(err.mean()) < 2e-3 * scale, (what, float(err.mean()), scale)   # and not merely inside the bound: typical error ~1e-3
    # the intermediate activations are fp16 and finite
    assert all(torch.isfinite(h.float()).all() for h in net.h)


@pytest.mark.gpu
def test_layer_kernel_exact_on_integer_data():
    """A = I-like / asymmetric small-integer operands: every product and sum is exact in fp16 / fp32, so the MFMA operand and
    accumulator lane maps are checked bit for bit (a transposed or permuted tile cannot pass)."""
    import torch
    from isaacgym_amd.policy import layer_forward
    m, n, k = 200, 150, 100           # ragged in every dimension
    gen = torch.Generator().manual_seed(1)
    a = torch.randint(-3, 4, (m, k), generator=gen).to(torch.float16)
    w = torch.randint(-3, 4, (n, k), generator=gen).to(torch.float16)
    w[:, 0] = torch.arange(n, dtype=torch.float16) % 5 - 2          # asymmetric
    bias = torch.randint(-2, 3, (n,), generator=gen).to(torch.float16)
    k8 = 104                                                          # rows padded to a multiple of 8 elements (16-byte rows)
    a_p, w_p = torch.zeros(m, k8, dtype=torch.float16), torch.zeros(n, k8, dtype=torch.float16)
    a_p[:, :k], w_p[:, :k] = a, w
    out = torch.zeros(m, n, dtype=torch.float32, device="cuda")
    layer_forward(out, a_p.cuda(), w_p.cuda(), bias.cuda(), elu=False, m=m, n=n, k=k)
    torch.cuda.synchronize()
    want = a.float() @ w.float().t() + bias.float()
    assert torch.equal(out.cpu(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [128, 384, 512, 513, 514, 515, 516, 517, 518, 520, 521])
@pytest.mark.parametrize("out_f32", [False, True])
def test_every_tile_exact_on_integer_data(tile, out_f32, monkeypatch):
    """The same bit-for-bit check for each tile configuration the launcher can pick (PPENV_MLP_TILE forces one), as the batched
    two-problem launch the hidden layers use: ragged M and N over several tiles, K = 192 (three K tiles of 64)."""
    import torch
    from isaacgym_amd.policy import layer_forward
    monkeypatch.setenv("PPENV_MLP_TILE", str(tile))
    m, n, k = 700, 600, 192
    gen = torch.Generator().manual_seed(tile)
    a = torch.randint(-3, 4, (m, 2 * k), generator=gen).to(torch.float16)
    w = torch.randint(-2, 3, (2, n, k), generator=gen).to(torch.float16)
    w[:, :, 0] = (torch.arange(n) % 5 - 2).to(torch.float16)
    bias = torch.randint(-2, 3, (2, n), generator=gen).to(torch.float16)
    out = torch.full((m, 2 * n), -7.0, dtype=torch.float32 if out_f32 else torch.float16, device="cuda")
    layer_forward(out, a.cuda(), w.cuda(), bias.cuda(), elu=False, batch=2, in_stride=k, w_stride=n * k, bias_stride=n, out_stride=n, m=m, n=n, k=k)
    torch.cuda.synchronize()
    want = torch.cat([a[:, j * k:(j + 1) * k].float() @ w[j].float().t() + bias[j].float() for j in range(2)], dim=1)
    assert float(want.abs().max()) < 2048                              # exact in fp16 too
    assert torch.equal(out.cpu().float(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [513, 514, 515, 518, 520, 521])
@pytest.mark.parametrize("k", [64, 128, 256, 320, 512])
def test_ring_kernels_exact_for_every_short_k(tile, k, monkeypatch):
    """The ring kernels keep S - 1 K tiles in flight (S = 5 for the 128 x 128 tile since round 4): K of one to eight tiles crosses every case
    of the prologue and of the tail's counted waits (fewer tiles than the ring is deep, exactly as many, more)."""
    import torch
    from isaacgym_amd.policy import layer_forward
    monkeypatch.setenv("PPENV_MLP_TILE", str(tile))
    m, n = 515, 384
    gen = torch.Generator().manual_seed(tile * 1000 + k)
    a = torch.randint(-3, 4, (m, 2 * k), generator=gen).to(torch.float16)
    w = torch.randint(-2, 3, (2, n, k), generator=gen).to(torch.float16)
    bias = torch.randint(-2, 3, (2, n), generator=gen).to(torch.float16)
    want = torch.cat([a[:, j * k:(j + 1) * k].float() @ w[j].float().t() + bias[j].float() for j in range(2)], dim=1)
    for rep in range(3):
        out = torch.full((m, 2 * n), -7.0, dtype=torch.float32, device="cuda")
        layer_forward(out, a.cuda(), w.cuda(), bias.cuda(), elu=False, batch=2, in_stride=k, w_stride=n * k, bias_stride=n, out_stride=n, m=m, n=n, k=k)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), want), (tile, k, rep)


@pytest.mark.gpu
def test_prepare_input_matches_torch():
    """ppenv_mlp_prepare_input: normalise, clamp, cast, zero-pad — bit for bit against the same fp32 arithmetic in torch."""
    import torch
    from isaacgym_amd.policy import prepare_input
    gen = torch.Generator().manual_seed(5)
    m, k, kp = 777, 313, 320
    obs = (torch.randn(m, k, generator=gen) * 3.0).cuda()
    mean, var = torch.randn(k, generator=gen).cuda(), (torch.rand(k, generator=gen) * 2 + 0.1).cuda()
    inv_std = torch.rsqrt(var + 1e-5)
    out = torch.full((m, kp), 9.0, dtype=torch.float16, device="cuda")
    prepare_input(out, obs, mean, inv_std, 5.0)
    want = torch.zeros(m, kp, dtype=torch.float16, device="cuda")
    want[:, :k] = torch.clamp((obs - mean) * inv_std, -5.0, 5.0).half()
    assert torch.equal(out, want)
    prepare_input(out, obs)                                         # no statistics: cast only
    want[:, :k] = obs.half()
    assert torch.equal(out, want)


@pytest.mark.gpu
def test_sample_actions_distribution_clamp_and_neglogp():
    """ppenv_mlp_sample_actions: Normal(mu, sigma) draws from the counter RNG (the reference samples with torch's generator: the stream
    is not pinnable, the distribution is), the clamp, the negative log-probability rl_games computes, determinism in (seed, counter)."""
    import torch
    from isaacgym_amd.policy import sample_actions
    m, a = 20000, 27
    gen = torch.Generator().manual_seed(3)
    head = torch.zeros(m, a + 1, device="cuda")                       # mu as NativeMLP hands it over: a view with row stride a + 1
    head[:, :a] = (torch.rand(m, a, generator=gen) - 0.5).cuda()
    mu = head[:, :a]
    sigma = (torch.rand(a, generator=gen) * 0.5 + 0.2).cuda()
    raw, nl = torch.zeros(m, a, device="cuda"), torch.zeros(m, device="cuda")
    sample_actions(raw, mu, sigma, 7, 1, 0.0, 0.0, nl)                  # lo >= hi: unclamped
    g = (raw - mu) / sigma
    assert abs(float(g.mean())) < 0.01 and abs(float(g.std()) - 1.0) < 0.01
    assert abs(float((g ** 3).mean())) < 0.03 and abs(float((g ** 4).mean()) - 3.0) < 0.1          # skewness 0, kurtosis 3
    cols = g.t() @ g / m                                                # independent across actions
    assert float((cols - torch.eye(a, device="cuda")).abs().max()) < 0.05
    want_nl = 0.5 * (g ** 2).sum(1) + torch.log(sigma).sum() + 0.5 * a * np.log(2 * np.pi)
    assert torch.allclose(nl, want_nl, rtol=1e-5, atol=1e-3)
    clamped = torch.zeros_like(raw)
    sample_actions(clamped, mu, sigma, 7, 1, -1.0, 1.0)
    assert torch.equal(clamped, raw.clamp(-1.0, 1.0))                   # same (seed, counter): same draws
    other = torch.zeros_like(raw)
    sample_actions(other, mu, sigma, 7, 2, 0.0, 0.0)
    assert float((other - raw).abs().mean()) > 0.1                      # next counter: new draws
    rows = (raw[1:] - mu[1:]) / sigma - (raw[:-1] - mu[:-1]) / sigma    # and rows are not copies of each other
    assert float(rows.abs().mean()) > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k", [(100, 28, 1024), (4096, 8, 512), (33, 32, 48)])
def test_heads_kernel_exact_on_integer_data(m, n, k):
    """The skinny fp32-output layer (mu | value heads: K split over the four waves of a workgroup, partial tiles summed through LDS)
    bit for bit on small-integer operands, ragged M, N <= 32, K steps that do not divide evenly among the waves."""
    import torch
    from isaacgym_amd.policy import layer_forward
    gen = torch.Generator().manual_seed(m + n)
    a = torch.randint(-3, 4, (m, k), generator=gen).to(torch.float16)
    w = torch.randint(-3, 4, (n, k), generator=gen).to(torch.float16)
    w[:, 1] = (torch.arange(n) % 7 - 3).to(torch.float16)
    bias = torch.randint(-2, 3, (n,), generator=gen).to(torch.float16)
    out = torch.full((m, n), -7.0, dtype=torch.float32, device="cuda")
    layer_forward(out, a.cuda(), w.cuda(), bias.cuda(), elu=False)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), a.float() @ w.float().t() + bias.float())


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [512, 513, 514, 515, 516, 517, 518, 520, 521])
def test_dma_tiles_exact_at_full_size_repeated(tile, monkeypatch):
    """Race screen for the LDS-DMA kernels' own barrier / vmcnt structure: the reference's widest layer (2048 -> 1536, two problems) at
    M = 4096 with the chip full, 32 K tiles per workgroup, small-integer operands so that every launch must reproduce the fp32 matmul
    bit for bit — a fragment read that beats its DMA, or a DMA that overwrites a tile still being read, shows up as a wrong element.
    Ten launches per configuration (tools/gpu_mlp_race_screen.py runs hundreds at several sizes)."""
    import torch
    from isaacgym_amd.policy import layer_forward
    monkeypatch.setenv("PPENV_MLP_TILE", str(tile))
    m, n, k = 4096, 1536, 2048
    gen = torch.Generator(device="cuda").manual_seed(tile)
    a = torch.randint(-1, 2, (m, 2 * k), generator=gen, device="cuda").to(torch.float16)
    w = torch.randint(-2, 3, (2, n, k), generator=gen, device="cuda").to(torch.float16)
    bias = torch.randint(-2, 3, (2, n), generator=gen, device="cuda").to(torch.float16)
    want = torch.cat([a[:, j * k:(j + 1) * k].float() @ w[j].float().t() + bias[j].float() for j in range(2)], dim=1)
    assert float(want.abs().max()) < 2 ** 24
    for rep in range(10):
        out = torch.full((m, 2 * n), -7.0, dtype=torch.float32, device="cuda")
        layer_forward(out, a, w, bias, elu=False, batch=2, in_stride=k, w_stride=n * k, bias_stride=n, out_stride=n, m=m, n=n, k=k)
        bad = int((out != want).sum())
        assert bad == 0, (tile, rep, bad)


@pytest.mark.gpu
@pytest.mark.parametrize("n,noise", [(4096, False), (1000, False), (1024, True)])
def test_step_kernel_writes_the_policy_input(n, noise):
    """SURVEY.md §8(f) N2, "fuse obs normalisation into the step kernel": with ppenv_ta_sim_set_policy_input the chain-wave step writes
    the policy's first-layer input next to obs_buf — bit for bit what ppenv_mlp_prepare_input makes of the obs_buf of the same step
    (ragged last workgroup included), obs_buf itself unchanged; and switched off again it is no longer written."""
    import torch
    from isaacgym_amd.policy import prepare_input
    from isaacgym_amd.tensor_api import TAEnv
    env = TAEnv(n, device="cuda:0", seed=3)
    if env.sim.kernel != "chain":
        pytest.skip("the chain-wave kernel is not the one in use")
    ref = TAEnv(n, device="cuda:0", seed=3)
    if noise:      # observation noise rewrites the whole tile after the task arithmetic: the policy's rows must be made from the NOISY row (what obs_buf holds)
        for e in (env, ref):
            e.set_randomization(action_noise_sigma=0.01, observation_noise_sigma=0.004)
    gen = torch.Generator(device="cuda").manual_seed(9)
    mean = torch.randn(313, device="cuda", generator=gen) * 0.3
    inv_std = torch.rand(313, device="cuda", generator=gen) * 3 + 0.2
    x16 = torch.full((n, 320), 9.0, dtype=torch.float16, device="cuda")
    env.set_policy_input(x16, mean, inv_std, 5.0)
    want = torch.empty_like(x16)
    for t in range(70):                                  # past episode ends: reset envs carry the observation of the restored state
        a = torch.rand(n, 27, device="cuda", generator=gen) * 2 - 1
        env.step(a)
        ref.step(a)
        assert torch.equal(env.obs_buf, ref.obs_buf), t
        prepare_input(want, env.obs_buf, mean, inv_std, 5.0)
        assert torch.equal(x16, want), t
    env.set_policy_input(None)
    x16.fill_(7.0)
    env.step(a)
    assert bool((x16 == 7.0).all())
    env.close(); ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("m,span", [(4096, (3, 6)), (4096, (5, 6)), (2048, (2, 5)), (1000, (3, 6)), (8192, (4, 6))])
def test_chained_layers_equal_the_per_layer_launches(m, span, monkeypatch):
    """ppenv_mlp_chain_forward: consecutive hidden layers of the reference's network as ONE launch (persistent workgroups, tiles by ticket, a
    tile waits for its rows of the layer below) — every activation buffer bit-identical to the per-layer launches (same tile kernels, same
    summation order), on random fp16 data with ELU, repeated (the release / acquire hand-off must hold launch after launch on the SAME buffers),
    ragged M included; the workspace's error word stays clear."""
    import torch
    from isaacgym_amd.policy import UNITS, _descriptor, chain_forward, chain_status, chain_workspace, layer_forward
    gen = torch.Generator(device="cuda").manual_seed(m + span[0])
    u = UNITS
    first, last = span                                     # 1-based hidden layers first .. last; layer i reads h[i - 2], writes h[i - 1]
    w = {i: (torch.randn(2, u[i - 1], u[i - 2], device="cuda", generator=gen) / u[i - 2] ** 0.5).half() for i in range(first, last + 1)}
    b = {i: (torch.randn(2, u[i - 1], device="cuda", generator=gen) * 0.1).half() for i in range(first, last + 1)}
    kw = lambda i, h: dict(out=h[i - 1], x=h[i - 2], w=w[i], bias=b[i], elu=True, batch=2, in_stride=u[i - 2], w_stride=u[i - 1] * u[i - 2], bias_stride=u[i - 1],
                           out_stride=u[i - 1], m=m, n=u[i - 1], k=u[i - 2])
    ws = chain_workspace(m, 2, last - first + 1, "cuda")
    for rep in range(6):
        x = torch.randn(m, 2 * u[first - 2], device="cuda", generator=gen).half()
        ha = {first - 2: x, **{i - 1: torch.full((m, 2 * u[i - 1]), 7.0, dtype=torch.float16, device="cuda") for i in range(first, last + 1)}}
        hb = {first - 2: x, **{i - 1: torch.full((m, 2 * u[i - 1]), -7.0, dtype=torch.float16, device="cuda") for i in range(first, last + 1)}} if rep == 0 else hb
        hb[first - 2] = x
        for i in range(first, last + 1):           # the per-layer launch on the tile the chain uses for this layer (128 x 256 where that gives 192 tiles, else 128 x 128)
            monkeypatch.setenv("PPENV_MLP_TILE", "521" if ((u[i - 1] + 255) // 256) * ((m + 127) // 128) * 2 >= 192 else "520")
            layer_forward(**kw(i, ha))
        monkeypatch.delenv("PPENV_MLP_TILE")
        chain_forward([_descriptor(**kw(i, hb)) for i in range(first, last + 1)], ws)
        torch.cuda.synchronize()
        for i in range(first, last + 1):
            assert torch.equal(ha[i - 1], hb[i - 1]), (rep, i, int((ha[i - 1] != hb[i - 1]).sum()))
    assert chain_status(ws) == 0
    assert int(ws[:2].abs().sum()) == 0 and int(ws[3:].abs().sum()) == 0          # the counters are left zeroed for the next launch


@pytest.mark.gpu
def test_attached_env_feeds_the_forward_from_the_first_call():
    """NativeMLP.attach_env: the step kernel writes the network's first-layer input, and attach_env itself fills it from the CURRENT obs_buf —
    forward(prepared=True) equals the forward with its own normalise-and-pad launch bit for bit before any step and after every step."""
    import torch
    from isaacgym_amd.policy import NativeMLP
    from isaacgym_amd.tensor_api import TAEnv
    n = 256
    env = TAEnv(n, device="cuda:0", seed=5)
    if env.sim.kernel != "chain":
        pytest.skip("the chain-wave kernel is not the one in use")
    gen = torch.Generator(device="cuda").manual_seed(2)
    dims = [313, 256, 128]
    mk = lambda n_out: [((torch.randn(o, i, device="cuda", generator=gen) / i ** 0.5), torch.randn(o, device="cuda", generator=gen) * 0.1)
                        for i, o in zip(dims, dims[1:] + [n_out])]
    actor, critic = mk(27), mk(1)
    kw = dict(mean=torch.randn(313, device="cuda", generator=gen) * 0.2, var=torch.rand(313, device="cuda", generator=gen) + 0.5, max_rows=n)
    fed, plain = NativeMLP(actor, critic, 313, "cuda:0", **kw), NativeMLP(actor, critic, 313, "cuda:0", **kw)
    fed.attach_env(env)
    for t in range(5):
        mu1, v1 = fed.forward(env.obs_buf, prepared=True)
        mu0, v0 = plain.forward(env.obs_buf)
        assert torch.equal(mu1, mu0) and torch.equal(v1, v0), t
        env.step(torch.clamp(mu0, -1, 1).contiguous())
    env.close()


def _rlgames_state_dict(torch, num_obs, units, num_act, gen):
    """A state dict with the key layout of rl_games' a2c_continuous_logstd model for a `separate: True` network."""
    sd, d = {}, num_obs
    for which in ("actor", "critic"):
        d = num_obs
        for li, u in enumerate(units):
            sd[f"a2c_network.{which}_mlp.{2 * li}.weight"] = (torch.rand(u, d, generator=gen) * 2 - 1) / np.sqrt(d)
            sd[f"a2c_network.{which}_mlp.{2 * li}.bias"] = (torch.rand(u, generator=gen) * 2 - 1) / np.sqrt(d)
            d = u
    sd["a2c_network.mu.weight"], sd["a2c_network.mu.bias"] = (torch.rand(num_act, d, generator=gen) * 2 - 1) / np.sqrt(d), torch.zeros(num_act)
    sd["a2c_network.value.weight"], sd["a2c_network.value.bias"] = (torch.rand(1, d, generator=gen) * 2 - 1) / np.sqrt(d), torch.zeros(1)
    sd["a2c_network.sigma"] = torch.full((num_act,), -2.0)
    sd["running_mean_std.running_mean"], sd["running_mean_std.running_var"] = torch.randn(num_obs, generator=gen) * 0.2, torch.rand(num_obs, generator=gen) + 0.5
    sd["running_mean_std.count"] = torch.tensor(1000.0)
    sd["value_mean_std.running_mean"], sd["value_mean_std.running_var"], sd["value_mean_std.count"] = torch.tensor([3.0]), torch.tensor([4.0]), torch.tensor(1000.0)
    return sd


def test_rlgames_state_dict_layout_is_parsed():
    """CPU: the key layout of an rl_games checkpoint's model -> ordered (weight, bias) lists, heads last; a shared-trunk network is refused."""
    import torch
    from isaacgym_amd.policy import layers_from_rlgames_state_dict
    sd = _rlgames_state_dict(torch, 80, (64, 32, 16), 7, torch.Generator().manual_seed(0))
    actor, critic = layers_from_rlgames_state_dict(sd)
    assert [tuple(w.shape) for w, _ in actor] == [(64, 80), (32, 64), (16, 32), (7, 16)]
    assert [tuple(w.shape) for w, _ in critic] == [(64, 80), (32, 64), (16, 32), (1, 16)]
    assert actor[1][0] is sd["a2c_network.actor_mlp.2.weight"] and critic[2][1] is sd["a2c_network.critic_mlp.4.bias"]
    with pytest.raises(KeyError):
        layers_from_rlgames_state_dict({k: v for k, v in sd.items() if "critic_mlp" not in k})


def test_rlgames_state_dict_layout_round_trips():
    """CPU: rlgames_state_dict_from_layers is the inverse of layers_from_rlgames_state_dict (keys, shapes, values), and the
    RunningMeanStd module saves / restores under rl_games' own buffer names, refreshing the fp32 tensors the kernels read IN PLACE."""
    import torch
    from isaacgym_amd.policy import RunningMeanStd, layers_from_rlgames_state_dict, rlgames_state_dict_from_layers
    sd = _rlgames_state_dict(torch, 80, (64, 32, 16), 7, torch.Generator().manual_seed(1))
    actor, critic = layers_from_rlgames_state_dict(sd)
    rms = RunningMeanStd(80, "cpu")
    assert sorted(rms.state_dict()) == ["count", "running_mean", "running_var"]          # mean / inv_std are derived, not saved
    held_mean, held_inv = rms.mean, rms.inv_std                                          # what NativeMLP.set_normalization_tensors keeps
    rms.load_state_dict({k.split(".", 1)[1]: v.double() for k, v in sd.items() if k.startswith("running_mean_std.")})
    assert held_mean is rms.mean and held_inv is rms.inv_std
    assert torch.allclose(held_mean, sd["running_mean_std.running_mean"].float())
    assert torch.allclose(held_inv, torch.rsqrt(sd["running_mean_std.running_var"].float() + 1e-5))
    back = rlgames_state_dict_from_layers(actor, critic, sigma=sd["a2c_network.sigma"], rms=rms)
    assert sorted(back) == sorted(k for k in sd if not k.startswith("value_mean_std."))
    for k, v in back.items():
        assert v.shape == sd[k].shape and torch.equal(v.to(sd[k].dtype), sd[k]), k
        assert v.data_ptr() != sd[k].data_ptr()                                         # clones


@pytest.mark.gpu
def test_actor_critic_checkpoint_round_trip_keeps_the_input_normaliser(tmp_path):
    """A policy trained with normalize_input, saved with model.state_dict() (what rl_games does), comes back with its statistics: the
    restored module, and the rl_games-layout export through from_rlgames and RLGamesPolicy, all reproduce mu / value exactly."""
    import torch
    from isaacgym_amd.policy import NativeActorCritic, RLGamesPolicy
    gen = torch.Generator().manual_seed(21)
    m, num_obs, num_act, units = 256, 80, 7, (256, 128)

    def mlp(n_out):
        layers, d = [], num_obs
        for u in units + (n_out,):
            layers.append((torch.randn(u, d, generator=gen) / d ** 0.5, torch.randn(u, generator=gen) * 0.1))
            d = u
        return layers
    net = NativeActorCritic(mlp(num_act), mlp(1), num_obs, "cuda:0", normalize_input=True)
    net.train()
    for _ in range(3):                                  # training-mode forwards move the statistics away from mean 0 / var 1
        net((torch.randn(m, num_obs, generator=gen) * 3.0 + 2.0).cuda())
    assert float(net.running_mean_std.running_mean.abs().min()) > 0.5 and float(net.running_mean_std.count) == 1 + 3 * m
    with torch.no_grad():
        net.sigma.fill_(-1.5)
    net.eval()
    obs = (torch.randn(m, num_obs, generator=gen) * 3.0 + 2.0).cuda()
    with torch.no_grad():
        mu, value = (t.clone() for t in net(obs))
    sd = net.state_dict()
    assert {"running_mean_std.running_mean", "running_mean_std.running_var", "running_mean_std.count"} <= set(sd)
    path = tmp_path / "native.pth"
    torch.save(sd, path)
    fresh = NativeActorCritic(mlp(num_act), mlp(1), num_obs, "cuda:0", normalize_input=True)      # other weights, statistics at their defaults
    fresh.load_state_dict(torch.load(path, weights_only=True))
    fresh.eval()
    with torch.no_grad():
        mu2, value2 = fresh(obs)
    assert torch.equal(mu2, mu) and torch.equal(value2, value)
    # ... and in rl_games' layout
    rl = net.to_rlgames_state_dict()
    assert rl["a2c_network.actor_mlp.2.weight"].shape == (128, 256) and rl["running_mean_std.count"].dtype == torch.float64
    again = NativeActorCritic.from_rlgames(rl, "cuda:0")
    again.eval()
    with torch.no_grad():
        mu3, value3 = again(obs)
    assert torch.equal(mu3, mu) and torch.equal(value3, value) and torch.equal(again.sigma, net.sigma)
    torch.save({"model": {k: v.cpu() for k, v in rl.items()}}, tmp_path / "rl.pth")
    pol = RLGamesPolicy.load(str(tmp_path / "rl.pth"), "cuda:0")
    act, val = pol.act(obs, deterministic=True)
    assert torch.equal(act, torch.clamp(mu, -1.0, 1.0)) and torch.equal(val, value)
    assert torch.allclose(pol.sigma, torch.full((num_act,), float(np.exp(-1.5)), device="cuda"))


@pytest.mark.gpu
def test_rlgames_checkpoint_is_served_by_the_native_forward(tmp_path):
    """A checkpoint file with rl_games' layout, loaded with weights_only, played deterministically: actions and de-normalised values
    against the same network evaluated in fp32 torch."""
    import torch
    from isaacgym_amd.policy import RLGamesPolicy
    gen = torch.Generator().manual_seed(4)
    units, num_obs, num_act, m = (2048, 1536, 1024, 1024, 512, 512), 313, 27, 1024
    sd = _rlgames_state_dict(torch, num_obs, units, num_act, gen)
    path = tmp_path / "last_ep_100.pth"
    torch.save({"model": sd, "epoch": 100, "frame": 1}, path)
    pol = RLGamesPolicy.load(str(path), "cuda:0")
    obs = torch.randn(m, num_obs, generator=gen) * 1.5
    act, val = pol.act(obs.cuda(), deterministic=True)
    x = torch.clamp((obs - sd["running_mean_std.running_mean"]) / torch.sqrt(sd["running_mean_std.running_var"] + 1e-5), -5, 5)

    def net(which, head):
        h = x
        for li in range(len(units)):
            h = torch.nn.functional.elu(h @ sd[f"a2c_network.{which}_mlp.{2 * li}.weight"].t() + sd[f"a2c_network.{which}_mlp.{2 * li}.bias"])
        return h @ sd[f"a2c_network.{head}.weight"].t() + sd[f"a2c_network.{head}.bias"]
    want_a, want_v = torch.clamp(net("actor", "mu"), -1, 1), net("critic", "value") * 2.0 + 3.0
    assert float((act.cpu() - want_a).abs().max()) < 2e-2 * max(1.0, float(want_a.abs().max()))
    assert float((val.cpu() - want_v).abs().max()) < 2e-2 * float(want_v.abs().max())
    assert torch.allclose(pol.sigma.cpu(), torch.full((num_act,), float(np.exp(-2.0))))
    drawn, _ = pol.act(obs.cuda(), deterministic=False)
    z = (drawn.cpu() - net("actor", "mu")) / float(np.exp(-2.0))
    inside = (drawn.cpu().abs() < 1.0)
    assert abs(float(z[inside].std()) - 1.0) < 0.05                   # Normal(mu, sigma) draws where the clamp did not bite


@pytest.mark.gpu
@pytest.mark.parametrize("m", [4096, 77])
def test_heads_sample_equals_heads_then_sample(m):
    """ppenv_mlp_heads_sample: the heads launch that also draws the actions writes the same mu / value, and bit for bit the same actions
    and log-probabilities, as the heads layer followed by ppenv_mlp_sample_actions with the same (seed, counter)."""
    import torch
    from isaacgym_amd.policy import heads_sample, layer_forward, sample_actions
    gen = torch.Generator().manual_seed(m)
    a, k = 27, 1024
    x = (torch.randn(m, k, generator=gen) * 0.5).half().cuda()
    w = (torch.randn(a + 1, k, generator=gen) / 32).half().cuda()
    b = (torch.randn(a + 1, generator=gen) * 0.1).half().cuda()
    sigma = (torch.rand(a, generator=gen) * 0.5 + 0.1).cuda()
    out1, out2 = torch.zeros(m, a + 1, device="cuda"), torch.zeros(m, a + 1, device="cuda")
    act1, act2 = torch.zeros(m, a, device="cuda"), torch.zeros(m, a, device="cuda")
    nl1, nl2 = torch.zeros(m, device="cuda"), torch.zeros(m, device="cuda")
    layer_forward(out1, x, w, b, elu=False)
    sample_actions(act1, out1[:, :a], sigma, 11, 5, -1.0, 1.0, nl1)
    heads_sample(out2, x, w, b, a, act2, sigma, 11, 5, -1.0, 1.0, nl2)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2) and torch.equal(act1, act2) and torch.equal(nl1, nl2)
    assert float(act1.abs().max()) <= 1.0 and float((act1 - out1[:, :a]).abs().mean()) > 0.05
