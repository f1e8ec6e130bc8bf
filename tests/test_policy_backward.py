"""N2, second half: the learner's backward of the policy MLP on the matrix cores (include/ppenv_policy.h: backward_input,
backward_weight, reduce_rows, bias_grad_f32, cast_weights, running_mean_std_update).

Parity: the whole backward against PyTorch autograd in fp32 on the same weights (fp16 operands, fp32 accumulation: rtol 1e-2 plus
1e-2 of each gradient's scale, as for the forward); every kernel alone bit-exact on small-integer operands (products and sums are
exact in fp16 / fp32, so a transposed, permuted or prematurely read tile cannot pass).  rl_games (the caller of these kernels in
the reference's training run, cfg/train/HumanoidPingpongTiltG1PPO.yaml) is not in the reference: the RunningMeanStd update is
checked against its published formula restated here in float64 torch — parity unpinned at that boundary."""
import ctypes as C

import numpy as np
import pytest

from test_policy_mlp import _mlp


def _ints(torch, gen, shape, lo=-2, hi=3):
    return torch.randint(lo, hi, shape, generator=gen).to(torch.float16)


# ------------------------------------------------------------------------------------------------ CPU: argument checks only
def test_backward_entry_points_reject_bad_arguments_before_any_device_call():
    from isaacgym_amd import _lib, policy
    L = policy._lib_policy()
    d = policy.MLPDw()
    assert L.ppenv_mlp_layer_backward_weight(C.byref(d), None) == -1 and b"NULL pointer" in L.ppenv_last_error()
    d.m, d.n, d.k, d.batch, d.lddz, d.ldx, d.lddw = 100, 64, 64, 1, 64, 64, 64            # m not a multiple of 64
    d.dz = d.x = d.dw = 4096
    assert L.ppenv_mlp_layer_backward_weight(C.byref(d), None) == -1 and b"m % 64" in L.ppenv_last_error()
    d.m, d.splits = 128, 3
    assert L.ppenv_mlp_layer_backward_weight(C.byref(d), None) == -1 and b"power of two" in L.ppenv_last_error()
    d.splits = 2                                                                           # needs a workspace
    assert L.ppenv_mlp_dw_workspace_bytes(C.byref(d)) == 2 * 64 * 64 * 4
    assert L.ppenv_mlp_layer_backward_weight(C.byref(d), None) == -1 and b"workspace" in L.ppenv_last_error()
    d.splits = 1
    assert L.ppenv_mlp_dw_workspace_bytes(C.byref(d)) == 0
    g = policy.MLPLayer()
    g.elu = 1
    assert L.ppenv_mlp_layer_backward_input(C.byref(g), None, 0, 0, None, 0, 0, None) == -1 and b"bias NULL, elu 0" in L.ppenv_last_error()
    assert L.ppenv_mlp_reduce_rows(None, 1, 1, 1, None, 0, None) == -1
    assert L.ppenv_mlp_cast_weights(None, 1, 1, 1, None, 0, None, 0, 0, None) == -1
    assert L.ppenv_mlp_cast_weights_batch(None, 0, None) == -1
    assert L.ppenv_running_mean_std_update(None, 1, 1, 1, None, None, None, None, None, 1e-5, None, None) == -1
    assert L.ppenv_running_mean_std_workspace_bytes(1000, 313) == 1024 + 8 * 2 * 313 * 8
    with pytest.raises(_lib.PPEnvError):
        _lib.check(-1)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k,batch,splits", [(64 * 6, 256, 256, 1, 1), (64 * 37, 328, 200, 1, 0), (64 * 37, 328, 200, 1, 4), (64 * 16, 72, 520, 2, 8),
                                                  (64 * 40, 264, 64, 2, 16), (64 * 5, 8, 512, 1, 1), (64 * 64, 512, 512, 2, 32), (64 * 3, 1032, 776, 1, 2)])
def test_weight_gradient_kernel_exact_on_integer_data(m, n, k, batch, splits):
    """dW = dZ^T X through the transposed LDS reads: ragged n / k (clamped chunks), several output tiles, every split count and XCD
    placement branch, batch strides, accumulate — bit for bit against the fp32 product, ten launches each."""
    import torch
    from isaacgym_amd.policy import layer_backward_weight
    gen = torch.Generator().manual_seed(m + n + k)
    dz, x = _ints(torch, gen, (m, batch * n)), _ints(torch, gen, (m, batch * k))
    want = torch.stack([dz[:, b * n:(b + 1) * n].float().t() @ x[:, b * k:(b + 1) * k].float() for b in range(batch)])
    assert float(want.abs().max()) < 2 ** 24
    dzd, xd = dz.cuda(), x.cuda()
    for rep in range(10):
        dw = torch.full((batch, n, k), float("nan"), device="cuda")
        layer_backward_weight(dw, dzd, xd, batch=batch, dz_stride=n, x_stride=k, dw_stride=n * k, n=n, k=k, splits=splits)
        torch.cuda.synchronize()
        assert torch.equal(dw.cpu(), want), (rep, float((dw.cpu() - want).abs().max()))
    base = torch.randint(-5, 5, (batch, n, k), generator=gen).float()
    dw = base.cuda()
    layer_backward_weight(dw, dzd, xd, batch=batch, dz_stride=n, x_stride=k, dw_stride=n * k, n=n, k=k, splits=splits, accumulate=True)
    assert torch.equal(dw.cpu(), want + base)
    # a padded gradient image (lddw > k, a batch stride that is not n k): the columns beyond k are left alone
    pad = torch.full((batch, n + 3, k + 8), 7.0, device="cuda")
    layer_backward_weight(pad[:, :n, :k], dzd, xd, batch=batch, dz_stride=n, x_stride=k, dw_stride=(n + 3) * (k + 8), n=n, k=k, splits=splits)
    assert torch.equal(pad[:, :n, :k].cpu(), want) and bool((pad[:, n:, :] == 7.0).all()) and bool((pad[:, :, k:] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [0, 128, 384, 512, 513, 514, 516, 520, 521])
@pytest.mark.parametrize("m,n,k,batch", [(333, 328, 192, 1), (640, 256, 128, 2), (1500, 136, 64, 2)])
def test_input_gradient_launch_exact_on_integer_data(monkeypatch, tile, m, n, k, batch):
    """dX = (dZ . Wt^T) * ELU'(y) and its per-64-row column sums on every tile kernel the launcher can pick (PPENV_MLP_TILE): exact
    operands, ELU outputs from {-0.5, -0.25, 0.5, 2} so the derivative {0.5, 0.75, 1, 1} is exact too."""
    import torch
    from isaacgym_amd.policy import layer_backward_input, reduce_rows
    if tile:
        monkeypatch.setenv("PPENV_MLP_TILE", str(tile))
    else:
        monkeypatch.delenv("PPENV_MLP_TILE", raising=False)
    gen = torch.Generator().manual_seed(7 * m + n)
    dz, wt = _ints(torch, gen, (m, batch * k)), _ints(torch, gen, (batch, n, k), -1, 2)
    y = torch.tensor([-0.5, -0.25, 0.5, 2.0])[torch.randint(0, 4, (m, batch * n), generator=gen)].to(torch.float16)
    want = torch.cat([dz[:, b * k:(b + 1) * k].float() @ wt[b].float().t() for b in range(batch)], dim=1) * torch.where(y.float() > 0, 1.0, y.float() + 1.0)
    assert float(want.abs().max()) < 2048                      # exact in fp16
    dx = torch.full((m, batch * n), float("nan"), dtype=torch.float16, device="cuda")
    blocks = (m + 63) // 64
    cs = torch.full((blocks, batch * n + 8), float("nan"), device="cuda")
    layer_backward_input(dx, dz.cuda(), wt.cuda(), elu_out=y.cuda(), colsum_partial=cs, batch=batch, dz_stride=k, wt_stride=n * k, dx_stride=n,
                         elu_out_stride=n, colsum_stride=n, m=m, n=n, k=k)
    torch.cuda.synchronize()
    assert torch.equal(dx.cpu().float(), want)
    pad = torch.zeros(blocks * 64 - m, batch * n)
    want_cs = torch.cat([want, pad]).view(blocks, 64, batch * n).sum(dim=1)
    assert torch.equal(cs.cpu()[:, :batch * n], want_cs)
    db = torch.empty(batch * n, device="cuda")
    reduce_rows(db, cs, rows=blocks, n=batch * n)
    assert torch.equal(db.cpu(), want.sum(dim=0))
    # without the ELU factor and without the sums: the plain product
    layer_backward_input(dx, dz.cuda(), wt.cuda(), batch=batch, dz_stride=k, wt_stride=n * k, dx_stride=n, m=m, n=n, k=k)
    plain = torch.cat([dz[:, b * k:(b + 1) * k].float() @ wt[b].float().t() for b in range(batch)], dim=1)
    assert torch.equal(dx.cpu().float(), plain)


@pytest.mark.gpu
def test_cast_weights_bias_grad_and_reduce_rows():
    import torch
    from isaacgym_amd.policy import bias_grad_f32, cast_weights, reduce_rows
    gen = torch.Generator().manual_seed(3)
    w = torch.randn(150, 313, generator=gen)
    w16 = torch.full((150, 320), float("nan"), dtype=torch.float16, device="cuda")
    wt16 = torch.full((320, 152), float("nan"), dtype=torch.float16, device="cuda")
    cast_weights(w.cuda(), w16, wt16)
    want = torch.zeros(150, 320, dtype=torch.float16)
    want[:, :313] = w.to(torch.float16)
    assert torch.equal(w16.cpu(), want)
    want_t = torch.zeros(320, 152, dtype=torch.float16)
    want_t[:313, :150] = w.to(torch.float16).t()
    assert torch.equal(wt16.cpu(), want_t)
    cast_weights(w.cuda(), None, wt16)                            # either image alone
    assert torch.equal(wt16.cpu(), want_t)
    # the whole-network form: several matrices (and a bias as a one-row matrix) in one launch
    from isaacgym_amd.policy import cast_item, cast_weights_batch
    w2, bias = torch.randn(70, 64, generator=gen).cuda(), torch.randn(1, 300, generator=gen).cuda()
    wd = w.cuda()
    o16, ot16 = torch.full((150, 320), float("nan"), dtype=torch.float16, device="cuda"), torch.full((320, 152), float("nan"), dtype=torch.float16, device="cuda")
    o2, b16 = torch.full((70, 64), float("nan"), dtype=torch.float16, device="cuda"), torch.full((1, 300), float("nan"), dtype=torch.float16, device="cuda")
    cast_weights_batch([cast_item(wd, o16, ot16), cast_item(w2, o2), cast_item(bias, b16)], wd)
    assert torch.equal(o16.cpu(), want) and torch.equal(ot16.cpu(), want_t)
    assert torch.equal(o2, w2.to(torch.float16)) and torch.equal(b16, bias.to(torch.float16))
    d = torch.randint(-8, 9, (5000, 28), generator=gen).float()
    out = torch.full((28,), 3.0, device="cuda")
    bias_grad_f32(out, d.cuda())
    assert torch.equal(out.cpu(), d.sum(dim=0))
    bias_grad_f32(out, d.cuda(), accumulate=True)
    assert torch.equal(out.cpu(), 2 * d.sum(dim=0))
    part = torch.randint(-8, 9, (37, 1001), generator=gen).float()
    dst = torch.ones(1001, device="cuda")
    reduce_rows(dst, part.cuda(), accumulate=True)
    assert torch.equal(dst.cpu(), part.sum(dim=0) + 1)


@pytest.mark.gpu
def test_running_mean_std_update_matches_the_rl_games_formula():
    """Three batches in a row (one not a multiple of the row block), then the statistics a NativeMLP reads."""
    import torch
    from isaacgym_amd.policy import RunningMeanStd
    gen = torch.Generator().manual_seed(5)
    k = 313
    rms = RunningMeanStd(k, "cuda:0")
    mean, var, count = torch.zeros(k, dtype=torch.float64), torch.ones(k, dtype=torch.float64), torch.ones((), dtype=torch.float64)
    for m in (4096, 1000, 32768):
        obs = torch.randn(m, k, generator=gen) * (torch.rand(k, generator=gen) * 5 + 0.1) + torch.randn(k, generator=gen) * 3
        rms.update(obs.cuda())
        x = obs.double()
        bm, bv = x.mean(0), x.var(0)                                # rl_games: input.mean(axis), input.var(axis) — unbiased
        delta, tot = bm - mean, count + m
        mean, var, count = (mean + delta * m / tot, (var * count + bv * m + delta ** 2 * count * m / tot) / tot, tot)
        torch.cuda.synchronize()
        np.testing.assert_allclose(rms.running_mean.cpu().numpy(), mean.numpy(), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(rms.running_var.cpu().numpy(), var.numpy(), rtol=1e-9, atol=1e-9)
        assert float(rms.count.cpu()) == float(count)
    np.testing.assert_allclose(rms.mean.cpu().numpy(), mean.float().numpy(), rtol=1e-6)
    np.testing.assert_allclose(rms.inv_std.cpu().numpy(), (1.0 / torch.sqrt(var.float() + 1e-5)).numpy(), rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("m,num_obs,num_act,units", [(4096, 313, 27, (2048, 1536, 1024, 1024, 512, 512)), (1024, 80, 7, (2048, 1536, 1024, 1024, 512, 512)),
                                                      (192, 80, 7, (256, 128))])
def test_native_backward_matches_fp32_autograd(m, num_obs, num_act, units):
    """forward + backward of the actor / critic pair against PyTorch autograd in fp32 on the same weights and the same d loss / d [mu |
    value]: every weight and bias gradient within rtol 1e-2 + 1e-2 of its scale (fp16 operands), mean error well inside."""
    import torch
    from isaacgym_amd.policy import NativeMLPLearner, RunningMeanStd
    gen = torch.Generator().manual_seed(m)
    actor, critic = _mlp(torch, num_obs, units, num_act, gen), _mlp(torch, num_obs, units, 1, gen)
    obs = torch.randn(m, num_obs, generator=gen) * 2.0 + 0.3
    learner = NativeMLPLearner(actor, critic, num_obs, "cuda:0")
    rms = RunningMeanStd(num_obs, "cuda:0")
    learner.attach_running_mean_std(rms)
    mu = learner.forward(obs.cuda(), update_stats=True)[0].clone()
    d_head = torch.randn(m, num_act + 1, generator=gen)           # O(1): what a loss scale makes of the 1 / M of a mean loss (fp16 gradients)
    d_head[:, num_act] *= 3.0
    grads = [g.clone() for g in learner.backward(d_head.cuda())]
    torch.cuda.synchronize()
    # the fp32 reference: the same normalisation (the statistics the update produced), autograd through both MLPs
    mean, inv_std = rms.mean.cpu(), rms.inv_std.cpu()
    x = torch.clamp((obs - mean) * inv_std, -5.0, 5.0)
    params = []

    def run(layers):
        h = x
        for i, (w, b) in enumerate(layers):
            w, b = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            params.append((w, b))
            h = h @ w.t() + b
            if i + 1 < len(layers):
                h = torch.nn.functional.elu(h)
        return h
    out = torch.cat([run(actor), run(critic)], dim=1)
    out.backward(d_head)
    nl = len(units)
    a_p, c_p = params[:nl + 1], params[nl + 1:]
    want = ([torch.stack([a_p[i][0].grad, c_p[i][0].grad]) for i in range(nl)] + [torch.stack([a_p[i][1].grad, c_p[i][1].grad]) for i in range(nl)] +
            [a_p[nl][0].grad, a_p[nl][1].grad, c_p[nl][0].grad, c_p[nl][1].grad])
    names = [f"w{i}" for i in range(nl)] + [f"b{i}" for i in range(nl)] + ["mu_w", "mu_b", "value_w", "value_b"]
    assert len(grads) == len(want) == len(learner.parameters())
    for name, g, w, p in zip(names, grads, want, learner.parameters()):
        g = g.cpu()
        assert g.shape == w.shape == p.shape, (name, g.shape, w.shape, p.shape)
        scale = float(w.abs().max())
        err = (g - w).abs()
        assert bool((err <= 1e-2 * w.abs() + 1e-2 * scale).all()), (name, float(err.max()), scale)
        assert float(err.mean()) < 2e-3 * scale, (name, float(err.mean()), scale)
    # accumulate = True adds a second copy
    grads2 = learner.backward(d_head.cuda(), accumulate=True)
    torch.cuda.synchronize()
    for name, g1, g2 in zip(names, grads, grads2):
        np.testing.assert_allclose(g2.cpu().numpy(), 2 * g1.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(g1.abs().max()), err_msg=name)   # (a + b) + partials: fp32 rounding
    # one SGD step on the masters, recast, forward again: the images follow the masters
    for p, g in zip(learner.parameters(), grads):
        p -= 0.5 * g
    learner.sync_weights()
    mu2, _ = learner.forward(obs.cuda())
    assert torch.isfinite(mu2).all() and not torch.equal(mu2, mu)


@pytest.mark.gpu
def test_learner_trains_with_a_torch_optimizer():
    """The pieces in a loop, as rl_games would drive them: forward (statistics learning), a loss on mu / value in PyTorch, the native backward,
    Adam on the fp32 masters, recast — the loss falls."""
    import torch
    from isaacgym_amd.policy import NativeMLPLearner, RunningMeanStd
    gen = torch.Generator().manual_seed(0)
    m, num_obs, num_act, units = 1024, 80, 7, (256, 128)
    actor, critic = _mlp(torch, num_obs, units, num_act, gen), _mlp(torch, num_obs, units, 1, gen)
    learner = NativeMLPLearner(actor, critic, num_obs, "cuda:0")
    learner.attach_running_mean_std(RunningMeanStd(num_obs, "cuda:0"))
    params = learner.parameters()
    opt = torch.optim.Adam(params, lr=1e-3)
    obs = (torch.randn(m, num_obs, generator=gen) * 3 + 1).cuda()
    target = torch.tanh(obs[:, :num_act + 1] * 0.3)                  # something learnable from the observations
    scale = 1024.0                                                   # a loss scale, as GradScaler would apply: fp16 gradients
    losses = []
    for it in range(60):
        mu, value = learner.forward(obs, update_stats=it < 5)
        out = torch.cat([mu, value], dim=1)
        losses.append(float(((out - target) ** 2).mean()))
        d_head = (2.0 * (out - target) / out.numel() * scale).contiguous()
        grads = learner.backward(d_head)
        for p, g in zip(params, grads):
            p.grad = (g / scale).contiguous()
        opt.step()
        learner.sync_weights()
    assert losses[-1] < 0.25 * losses[0], (losses[0], losses[-1])
    assert all(torch.isfinite(p).all() for p in params)


@pytest.mark.gpu
def test_native_actor_critic_module_is_an_ordinary_differentiable_torch_module():
    """NativeActorCritic: loss.backward() fills .grad through the native backward (against fp32 autograd on the same weights), a torch
    optimizer's step is picked up by the next forward (recast on parameter-version change), eval mode leaves the statistics alone."""
    import torch
    from isaacgym_amd.policy import NativeActorCritic
    gen = torch.Generator().manual_seed(11)
    m, num_obs, num_act, units = 256, 80, 7, (256, 128)
    actor, critic = _mlp(torch, num_obs, units, num_act, gen), _mlp(torch, num_obs, units, 1, gen)
    net = NativeActorCritic(actor, critic, num_obs, "cuda:0", normalize_input=False)
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) == sum(w.numel() + b.numel() for w, b in actor + critic)
    obs = torch.randn(m, num_obs, generator=gen)
    adv = torch.randn(m, generator=gen)
    ret = torch.randn(m, 1, generator=gen)
    act = torch.randn(m, num_act, generator=gen)

    def loss_of(mu, value):                          # a PPO-shaped loss: Gaussian log-probability weighted by an advantage + a value loss
        return (((act.to(mu.device) - mu) ** 2).sum(dim=1) * adv.to(mu.device)).mean() * 0.5 + ((value - ret.to(mu.device)) ** 2).mean()
    mu, value = net(obs.cuda())
    assert mu.requires_grad and value.requires_grad
    loss_of(mu, value).backward()
    ref = []

    def run(layers):
        h = torch.clamp(obs, -5.0, 5.0)
        for i, (w, b) in enumerate(layers):
            w, b = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            ref.append((w, b))
            h = h @ w.t() + b
            if i + 1 < len(layers):
                h = torch.nn.functional.elu(h)
        return h
    loss_of(run(actor), run(critic)).backward()
    nl = len(units)
    for i in range(nl):
        for j, off in ((0, 0), (1, nl + 1)):
            for got, want in ((net.hidden_w[i].grad[j], ref[off + i][0].grad), (net.hidden_b[i].grad[j], ref[off + i][1].grad)):
                scale = float(want.abs().max())
                assert float((got.cpu() - want).abs().max()) <= 2e-2 * scale, (i, j, scale)
    for got, want in ((net.mu_w.grad, ref[nl][0].grad), (net.mu_b.grad, ref[nl][1].grad), (net.value_w.grad, ref[2 * nl + 1][0].grad), (net.value_b.grad, ref[2 * nl + 1][1].grad)):
        assert float((got.cpu() - want).abs().max()) <= 2e-2 * float(want.abs().max())
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    before = mu.detach().clone()
    opt.step()
    mu2, _ = net(obs.cuda())                          # the step changed the masters: the fp16 images were recast
    assert not torch.allclose(mu2.detach(), before)
    with pytest.raises(ValueError, match="multiple of 64"):
        net(obs[:100].cuda())
    net.eval()
    with torch.no_grad():
        mu3, _ = net(obs.cuda())
    assert torch.equal(mu3, mu2.detach())


@pytest.mark.gpu
def test_backward_refuses_a_gradient_of_an_overwritten_forward():
    """The activation buffers are the learner's own: a second forward before the first loss.backward() used to give silently wrong
    gradients; now the backward of the stale graph raises, and accumulate + on_grads (double-reduced sums) is refused too."""
    import torch
    from isaacgym_amd.policy import NativeActorCritic
    gen = torch.Generator().manual_seed(12)
    m, num_obs, num_act, units = 128, 80, 7, (128, 64)
    net = NativeActorCritic(_mlp(torch, num_obs, units, num_act, gen), _mlp(torch, num_obs, units, 1, gen), num_obs, "cuda:0", normalize_input=False)
    x1, x2 = torch.randn(m, num_obs, generator=gen).cuda(), torch.randn(m, num_obs, generator=gen).cuda()
    mu_old, _ = net(x1)                               # e.g. the old policy's evaluation ...
    mu_new, v_new = net(x2)                           # ... then the new one's, same row count
    with pytest.raises(RuntimeError, match="forward #1, but the activation buffers hold forward #2"):
        mu_old.sum().backward()
    (mu_new.sum() + v_new.sum()).backward()           # the latest forward's backward is the valid one
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters() if p.requires_grad)
    lr = net.learner
    d_head = torch.ones(m, num_act + 1, device="cuda")
    with pytest.raises(ValueError, match="accumulate"):
        lr.backward(d_head, accumulate=True, on_grads=lambda name, tensors: None)


@pytest.mark.gpu
def test_learner_forward_backward_is_graph_capturable():
    """The minibatch step's native part only enqueues (no allocation, no synchronisation once its buffers exist): RunningMeanStd update + forward +
    backward captured into one HIP graph; a replay reproduces the eager gradients bit for bit."""
    import torch
    from isaacgym_amd.policy import NativeMLPLearner, RunningMeanStd
    gen = torch.Generator().manual_seed(2)
    m, num_obs, num_act, units = 512, 80, 7, (256, 128)
    actor, critic = _mlp(torch, num_obs, units, num_act, gen), _mlp(torch, num_obs, units, 1, gen)
    learner = NativeMLPLearner(actor, critic, num_obs, "cuda:0")
    learner.attach_running_mean_std(RunningMeanStd(num_obs, "cuda:0"))
    obs = torch.randn(m, num_obs, generator=gen).cuda()
    d_head = torch.randn(m, num_act + 1, generator=gen).cuda()
    with torch.no_grad():
        learner.forward(obs, update_stats=True)
        eager = [g.clone() for g in learner.backward(d_head)]              # also allocates every buffer
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            learner.forward(obs)
            learner.backward(d_head)
        for t in learner.gradients():
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
    for a, b in zip(eager, learner.gradients()):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_learner_backward_all_reduces_each_layer_on_rccl_with_one_rank():
    """The data-parallel learners' collective on the real transport, as far as a one-GPU box allows: NativeActorCritic with
    grad_sync = GradientBuckets(force=True) in a one-rank nccl group — every layer's gradients go through an asynchronous all_reduce issued right after
    the launches that produce them (heads first, layer 1 last), and come out as they went in (a mean over one rank)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from isaacgym_amd import distributed as D
from isaacgym_amd.policy import NativeActorCritic
from test_policy_mlp import _mlp
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
gen = torch.Generator().manual_seed(4)
actor, critic = _mlp(torch, 80, (256, 128), 7, gen), _mlp(torch, 80, (256, 128), 1, gen)
obs, w = torch.randn(512, 80, generator=gen).cuda(), torch.randn(512, 8, generator=gen).cuda()
grads = []
for sync in (None, D.GradientBuckets(force=True)):
    net = NativeActorCritic(actor, critic, 80, dev, normalize_input=False)
    net.grad_sync = sync
    mu, value = net(obs)
    (torch.cat([mu, value], dim=1) * w).sum().backward()
    torch.cuda.synchronize()
    grads.append([p.grad.clone() for p in net.parameters() if p.requires_grad])
assert sync.active and sync.names == ["heads", "layer2", "layer1"], sync.names
assert sync.bytes == sum(t.numel() * 4 for t in net.learner.grads["w"] + net.learner.grads["b"]) + net.learner.grads["head_w"].numel() * 4 + net.learner.grads["head_b"].numel() * 4
for a, b in zip(*grads):
    assert torch.equal(a, b)
dist.barrier(); dist.destroy_process_group(); print("ok")
''' % (root, root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]
