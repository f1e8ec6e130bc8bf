"""The policy forward of the rollout loop on the matrix cores (C ABI: include/ppenv_policy.h).

rl_games' a2c_continuous network of the reference (cfg/train/HumanoidPingpongTiltG1PPO.yaml:10-31,50-51): separate actor and critic
MLPs, units [2048, 1536, 1024, 1024, 512, 512], ELU, a linear mu head (fixed sigma) and a linear value head, inputs normalised by a
RunningMeanStd (clamped to +-5), mixed precision.  `NativeMLP.forward(obs_buf)` runs it as eight launches — a normalise-and-pad pass, then seven of the hand-written MFMA
kernel (v_mfma_f32_32x32x16_f16, fp32 accumulation):

    input       obs_buf [M, num_obs] fp32 -> normalised, clamped fp16, K padded to a multiple of 64 (or fused into layer 1: fuse_input)
    layer 1     actor | critic as one N = 4096 GEMM
    layers 2-6  actor and critic as the two problems of one batched launch, bias + ELU on the accumulators, fp16 activations
    heads       mu [M, num_actions] and value [M, 1] in fp32, one block-diagonal layer over [actor | critic] features

PyTorch is only the owner of the device buffers here.  Weights are cast to fp16 once (`load`), as autocast does per call.
"""
import ctypes as C

import torch

from . import _lib

UNITS = [2048, 1536, 1024, 1024, 512, 512]     # cfg/train/HumanoidPingpongTiltG1PPO.yaml:29


class MLPLayer(C.Structure):
    """ctypes mirror of ppenv_mlp_layer (include/ppenv_policy.h)."""
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("batch", C.c_int32),
                ("in_", C.c_void_p), ("in_stride", C.c_int64), ("lda", C.c_int32), ("in_f32", C.c_int32),
                ("mean", C.c_void_p), ("inv_std", C.c_void_p), ("clip", C.c_float),
                ("w", C.c_void_p), ("w_stride", C.c_int64), ("ldw", C.c_int32),
                ("bias", C.c_void_p), ("bias_stride", C.c_int64), ("elu", C.c_int32),
                ("out", C.c_void_p), ("out_stride", C.c_int64), ("ldo", C.c_int32), ("out_f32", C.c_int32)]


def _lib_policy():
    L = _lib.lib()
    L.ppenv_mlp_layer_forward.argtypes = [C.POINTER(MLPLayer), C.c_void_p]
    L.ppenv_mlp_prepare_input.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]
    L.ppenv_gae.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ppenv_mlp_heads_sample.argtypes = [C.POINTER(MLPLayer), C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ppenv_mlp_sample_actions.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def sample_actions(actions, mu, sigma, seed, counter, lo=-1.0, hi=1.0, neglogp=None):
    """ppenv_mlp_sample_actions on torch tensors: actions [M, A] = clamp(mu + sigma * N(0, 1), lo, hi), neglogp [M] of the unclamped
    draw (rl_games a2c_continuous with fixed sigma); deterministic in (seed, counter)."""
    L = _lib_policy()
    assert actions.is_contiguous() and mu.stride(1) == 1 and sigma.is_contiguous()
    _lib.check(L.ppenv_mlp_sample_actions(mu.data_ptr(), mu.shape[0], mu.shape[1], mu.stride(0), sigma.data_ptr(), seed, counter, lo, hi,
                                          actions.data_ptr(), neglogp.data_ptr() if neglogp is not None else None,
                                          torch.cuda.current_stream(mu.device).cuda_stream))


def prepare_input(out, obs, mean=None, inv_std=None, clip=5.0):
    """ppenv_mlp_prepare_input on torch tensors: obs fp32 [M, K] -> out fp16 [M, Kpad] normalised, clamped, zero-padded."""
    L = _lib_policy()
    _lib.check(L.ppenv_mlp_prepare_input(obs.data_ptr(), obs.shape[0], obs.shape[1], obs.stride(0), mean.data_ptr() if mean is not None else None,
                                         inv_std.data_ptr() if inv_std is not None else None, clip, out.data_ptr(), out.stride(0),
                                         torch.cuda.current_stream(obs.device).cuda_stream))


def _descriptor(out, x, w, bias, elu, batch=1, in_stride=0, w_stride=0, bias_stride=0, out_stride=0, mean=None, inv_std=None, clip=5.0,
                m=None, n=None, k=None):
    d = MLPLayer()
    d.m = x.shape[0] if m is None else m
    d.n = (w.shape[-2] if n is None else n)
    d.k = (w.shape[-1] if k is None else k)
    d.batch = batch
    d.in_, d.in_stride, d.lda, d.in_f32 = x.data_ptr(), in_stride, x.stride(0), int(x.dtype == torch.float32)
    d.mean = mean.data_ptr() if mean is not None else None
    d.inv_std = inv_std.data_ptr() if inv_std is not None else None
    d.clip = clip
    d.w, d.w_stride, d.ldw = w.data_ptr(), w_stride, w.stride(-2)
    d.bias, d.bias_stride = (bias.data_ptr() if bias is not None else None), bias_stride
    d.elu = int(elu)
    d.out, d.out_stride, d.ldo, d.out_f32 = out.data_ptr(), out_stride, out.stride(0), int(out.dtype == torch.float32)
    return d


def layer_forward(out, x, w, bias, elu, **kw):
    """One launch of ppenv_mlp_layer_forward on torch tensors (x: fp16 activations or fp32 observations; w: fp16 [n, k])."""
    d = _descriptor(out, x, w, bias, elu, **kw)
    _lib.check(_lib_policy().ppenv_mlp_layer_forward(C.byref(d), torch.cuda.current_stream(x.device).cuda_stream))


def heads_sample(out, x, w, bias, num_actions, actions, sigma, seed, counter, lo=-1.0, hi=1.0, neglogp=None):
    """ppenv_mlp_heads_sample: the heads layer out [M, n <= 32] (fp32) = x . w^T + bias and, in the same launch, what
    sample_actions(actions, out[:, :num_actions], sigma, seed, counter, lo, hi, neglogp) would produce."""
    d = _descriptor(out, x, w, bias, False)
    assert actions.is_contiguous() and actions.shape[1] == num_actions and sigma.is_contiguous()
    _lib.check(_lib_policy().ppenv_mlp_heads_sample(C.byref(d), num_actions, sigma.data_ptr(), seed, counter, lo, hi, actions.data_ptr(),
                                                    neglogp.data_ptr() if neglogp is not None else None,
                                                    torch.cuda.current_stream(x.device).cuda_stream))


class NativeMLP:
    """Actor + critic forward on the MFMA kernel.  `actor` / `critic`: lists of (weight [out, in], bias [out]) fp32 tensors, hidden
    layers first, the head last (what `[m for m in net if isinstance(m, nn.Linear)]` yields for the reference's architecture)."""

    def __init__(self, actor, critic, num_obs, device, mean=None, var=None, eps=1e-5, clip=5.0, max_rows=None, fuse_input=False):
        """fuse_input: layer 1 reads the fp32 observations in place and normalises while staging (one launch fewer, but the
        register-staged kernel); default: a small normalise-and-pad launch first, then layer 1 on the LDS-DMA kernel like the rest
        (M = 4096, 313 observations: 54 us fused, see DESIGN.md §5a for the split path)."""
        self.device = torch.device(device)
        self.fuse_input = bool(fuse_input)
        assert len(actor) == len(critic) and all(a[0].shape[0] == c[0].shape[0] for a, c in zip(actor[:-1], critic[:-1]))
        self.num_obs, self.clip = int(num_obs), float(clip)
        self.units = [a[0].shape[0] for a in actor[:-1]]
        self.num_actions = actor[-1][0].shape[0]
        self.load(actor, critic)
        self.set_normalization(mean, var, eps)
        self._rows = 0
        if max_rows:
            self._alloc(max_rows)

    def load(self, actor, critic):
        """fp32 master weights -> the fp16 operand images (actor | critic stacked per layer)."""
        h = lambda t: t.detach().to(self.device, torch.float16).contiguous()

        def hw(t):   # weight rows zero-padded to a multiple of 64 fp16 (one K tile): every row starts 16-byte aligned and layer 1
            t = h(t)   # qualifies for the LDS-DMA kernels (num_obs = 313 gives 626-byte rows; unpadded, it fell back to element loads)
            k = t.shape[1]
            kp = (k + 63) // 64 * 64
            if kp == k:
                return t
            out = torch.zeros((t.shape[0], kp), dtype=torch.float16, device=self.device)
            out[:, :k] = t
            return out
        self.w, self.b = [], []
        for (wa, ba), (wc, bc) in zip(actor[:-1], critic[:-1]):
            self.w.append(torch.stack([hw(wa), hw(wc)]).contiguous())     # [2, n, k (padded)]
            self.b.append(torch.stack([h(ba), h(bc)]).contiguous())       # [2, n]
        # the two heads as ONE block-diagonal layer over the stacked features [actor | critic]: rows 0 .. A-1 read the actor half, row A the critic half
        ul, na = self.units[-1], actor[-1][0].shape[0]
        self.head_w = torch.zeros((na + 1, 2 * ul), dtype=torch.float16, device=self.device)
        self.head_w[:na, :ul] = h(actor[-1][0])
        self.head_w[na:, ul:] = h(critic[-1][0])
        self.head_b = torch.cat([h(actor[-1][1]), h(critic[-1][1])]).contiguous()

    def set_normalization(self, mean, var, eps=1e-5):
        """rl_games RunningMeanStd in eval mode: (x - mean) / sqrt(var + eps), then clamp(+-clip)."""
        if mean is None:
            self.mean = self.inv_std = None
        else:
            self.mean = mean.detach().to(self.device, torch.float32).contiguous()
            self.inv_std = torch.rsqrt(var.detach().to(self.device, torch.float32) + eps).contiguous()

    def _alloc(self, m):
        z = lambda n, dt: torch.empty((m, n), dtype=dt, device=self.device)
        self.h = [z(2 * u, torch.float16) for u in self.units]            # actor columns first, critic after
        self.head_out = z(self.num_actions + 1, torch.float32)
        self.mu, self.value = self.head_out[:, :self.num_actions], self.head_out[:, self.num_actions:]       # views: [M, A] and [M, 1]
        self.x16 = None if self.fuse_input else z(self.w[0].shape[-1], torch.float16)   # normalised observations, K padded like the weights
        self._rows = m

    def attach_env(self, env):
        """Let the env's step kernel write this network's first-layer input (normalised, clamped, padded fp16 rows) next to obs_buf:
        `forward(obs, prepared=True)` then skips the normalise-and-pad launch.  Needs an env with `set_policy_input` (TAEnv on the
        chain-wave kernel) and statistics (`set_normalization`); the statistics tensors are read by every later step."""
        assert not self.fuse_input and self.mean is not None
        if env.num_envs != self._rows:
            self._alloc(env.num_envs)
        env.set_policy_input(self.x16, self.mean, self.inv_std, self.clip)

    def forward(self, obs, prepared=False, sample=None, head_out=None):
        """obs: fp32 [M, num_obs] on this device (the env's obs_buf, read in place) -> (mu [M, A], value [M, 1]) fp32 (buffers reused).
        prepared: the first-layer input is already in self.x16 (attach_env: written by the env's step kernel).
        sample: dict(actions=[M, A] fp32, sigma=[A], seed=, counter=, lo=-1, hi=1, neglogp=[M] or None) — the heads launch also draws the
        actions (ppenv_mlp_heads_sample): one launch fewer than forward + sample_actions, the same numbers.
        head_out: [M, A + 1] fp32, row stride A + 1 — where mu | value go instead of the network's own buffer (a rollout collector's
        horizon-major slice); the returned views then point into it."""
        m = obs.shape[0]
        if m != self._rows:
            assert not prepared
            self._alloc(m)
        assert obs.dtype == torch.float32 and obs.device == self.device and obs.stride(1) == 1 and obs.shape[1] == self.num_obs
        u = self.units
        # layer 1: both networks read the same rows -> one N = 2 u0 GEMM over the stacked weights
        w0 = self.w[0].view(2 * u[0], self.w[0].shape[-1])
        if self.fuse_input:
            layer_forward(self.h[0], obs, w0, self.b[0].view(-1), elu=True, mean=self.mean, inv_std=self.inv_std, clip=self.clip, k=self.num_obs)
        else:
            if not prepared:
                prepare_input(self.x16, obs, self.mean, self.inv_std, self.clip)
            layer_forward(self.h[0], self.x16, w0, self.b[0].view(-1), elu=True)
        for i in range(1, len(u)):
            layer_forward(self.h[i], self.h[i - 1], self.w[i], self.b[i], elu=True, batch=2, in_stride=u[i - 1], w_stride=u[i] * self.w[i].shape[-1],
                          bias_stride=u[i], out_stride=u[i], m=m, n=u[i], k=u[i - 1])
        ho = self.head_out if head_out is None else head_out
        mu, value = ho[:, :self.num_actions], ho[:, self.num_actions:]
        if sample is None or self.num_actions + 1 > 32:        # the skinny heads kernel (and with it the fused draw) takes up to 32 columns
            layer_forward(ho, self.h[-1], self.head_w, self.head_b, elu=False)
            if sample is not None:
                sample_actions(sample["actions"], mu, sample["sigma"], sample["seed"], sample["counter"], sample.get("lo", -1.0), sample.get("hi", 1.0),
                               sample.get("neglogp"))
        else:
            heads_sample(ho, self.h[-1], self.head_w, self.head_b, self.num_actions, sample["actions"], sample["sigma"], sample["seed"],
                         sample["counter"], sample.get("lo", -1.0), sample.get("hi", 1.0), sample.get("neglogp"))
        return mu, value

    @staticmethod
    def flops(m, num_obs, units=UNITS, num_actions=0):
        dims = [num_obs] + list(units)
        per_net = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        return 2 * m * (2 * per_net + units[-1] * (num_actions + 1))


# ---- a trained rl_games checkpoint on the native forward (the reference's `train.py test=True checkpoint=...` play mode) ---------------
_MLP_KEY = r"a2c_network\.%s_mlp\.(\d+)\.weight"


def layers_from_rlgames_state_dict(sd):
    """rl_games' a2c_continuous(_logstd) model state_dict (what `torch.load(ckpt)["model"]` holds for the reference's network:
    `separate: True`, cfg/train/HumanoidPingpongTiltG1PPO.yaml:10-31) -> (actor, critic) lists of (weight, bias), head last.
    Keys: a2c_network.{actor,critic}_mlp.<even index>.{weight,bias} (nn.Sequential of Linear / activation), a2c_network.mu.*,
    a2c_network.value.*."""
    import re

    def mlp(which):
        idx = sorted(int(m.group(1)) for k in sd for m in [re.fullmatch(_MLP_KEY % which, k)] if m)
        if not idx:
            raise KeyError(f"no a2c_network.{which}_mlp.*.weight in the state dict (a `separate: True` a2c network is expected)")
        return [(sd[f"a2c_network.{which}_mlp.{i}.weight"], sd[f"a2c_network.{which}_mlp.{i}.bias"]) for i in idx]
    actor = mlp("actor") + [(sd["a2c_network.mu.weight"], sd["a2c_network.mu.bias"])]
    critic = mlp("critic") + [(sd["a2c_network.value.weight"], sd["a2c_network.value.bias"])]
    return actor, critic


class RLGamesPolicy:
    """A trained rl_games continuous-action policy served by NativeMLP: input normalisation (running_mean_std, eval mode), actor and
    critic, fixed log-std sigma (`continuous_a2c_logstd`: sigma = exp(a2c_network.sigma)), value de-normalisation (value_mean_std,
    `normalize_value: True`).  `act(obs)` is what rl_games' player does per step: mu (deterministic) or a Normal(mu, sigma) draw,
    clamped to [-1, 1]."""

    def __init__(self, state_dict, device, max_rows=None, eps=1e-5):
        sd = state_dict
        actor, critic = layers_from_rlgames_state_dict(sd)
        num_obs = actor[0][0].shape[1]
        mean, var = sd.get("running_mean_std.running_mean"), sd.get("running_mean_std.running_var")
        if mean is None:                                       # normalize_input: False
            mean, var = torch.zeros(num_obs), torch.ones(num_obs) - eps
        self.net = NativeMLP(actor, critic, num_obs, device, mean=mean.float(), var=var.float(), eps=eps, max_rows=max_rows)
        self.device = self.net.device
        self.sigma = torch.exp(sd["a2c_network.sigma"].detach().float()).to(self.device).contiguous()
        vm, vv = sd.get("value_mean_std.running_mean"), sd.get("value_mean_std.running_var")
        self.value_mean = float(vm.reshape(-1)[0]) if vm is not None else 0.0
        self.value_std = float(torch.sqrt(vv.reshape(-1)[0].float() + eps)) if vv is not None else 1.0
        self._actions = self._neglogp = None
        self._counter = 0

    @classmethod
    def load(cls, path, device, max_rows=None):
        """A checkpoint file written by rl_games (`nn/<name>.pth`): tensors only are read (weights_only)."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        return cls(ckpt["model"] if "model" in ckpt else ckpt, device, max_rows=max_rows)

    def act(self, obs, deterministic=True, seed=0):
        """obs [M, num_obs] fp32 on the device -> (actions [M, A] in [-1, 1], value [M, 1] de-normalised)."""
        m = obs.shape[0]
        if self._actions is None or self._actions.shape[0] != m:
            self._actions = torch.empty(m, self.net.num_actions, device=self.device)
            self._neglogp = torch.empty(m, device=self.device)
        if deterministic:
            mu, v = self.net.forward(obs)
            torch.clamp(mu, -1.0, 1.0, out=self._actions)
        else:                                                  # the heads launch draws the actions as well
            self._counter += 1
            mu, v = self.net.forward(obs, sample=dict(actions=self._actions, sigma=self.sigma, seed=seed, counter=self._counter, neglogp=self._neglogp))
        return self._actions, v * self.value_std + self.value_mean
