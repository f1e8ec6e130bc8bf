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
import os

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


class MLPDw(C.Structure):
    """ctypes mirror of ppenv_mlp_dw (include/ppenv_policy.h)."""
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("batch", C.c_int32),
                ("dz", C.c_void_p), ("dz_stride", C.c_int64), ("lddz", C.c_int32),
                ("x", C.c_void_p), ("x_stride", C.c_int64), ("ldx", C.c_int32),
                ("dw", C.c_void_p), ("dw_stride", C.c_int64), ("lddw", C.c_int32),
                ("accumulate", C.c_int32), ("splits", C.c_int32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class MLPCast(C.Structure):
    """ctypes mirror of ppenv_mlp_cast (include/ppenv_policy.h)."""
    _fields_ = [("w32", C.c_void_p), ("n", C.c_int32), ("k", C.c_int32), ("ldw32", C.c_int32),
                ("w16", C.c_void_p), ("ldw16", C.c_int32),
                ("wt16", C.c_void_p), ("ldwt16", C.c_int32), ("wt_rows", C.c_int32)]


def _lib_policy():
    L = _lib.lib()
    if getattr(L, "_policy_bound", False):
        return L
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ppenv_mlp_layer_backward_input.argtypes = [C.POINTER(MLPLayer), vp, i64, i32, vp, i64, i32, vp]
    L.ppenv_mlp_dw_workspace_bytes.restype = C.c_size_t
    L.ppenv_mlp_dw_workspace_bytes.argtypes = [C.POINTER(MLPDw)]
    L.ppenv_mlp_layer_backward_weight.argtypes = [C.POINTER(MLPDw), vp]
    L.ppenv_mlp_reduce_rows.argtypes = [vp, i32, i64, i64, vp, i32, vp]
    L.ppenv_mlp_bias_grad_workspace_bytes.restype = C.c_size_t
    L.ppenv_mlp_bias_grad_workspace_bytes.argtypes = [i32, i32]
    L.ppenv_mlp_bias_grad_f32.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp]
    L.ppenv_mlp_cast_weights.argtypes = [vp, i32, i32, i32, vp, i32, vp, i32, i32, vp]
    L.ppenv_mlp_cast_weights_batch.argtypes = [C.POINTER(MLPCast), i32, vp]
    L.ppenv_running_mean_std_workspace_bytes.restype = C.c_size_t
    L.ppenv_running_mean_std_workspace_bytes.argtypes = [i32, i32]
    L.ppenv_running_mean_std_update.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, C.c_float, vp, vp]
    L._policy_bound = True
    L.ppenv_mlp_layer_forward.argtypes = [C.POINTER(MLPLayer), C.c_void_p]
    L.ppenv_mlp_layer_forward_share.argtypes = [C.POINTER(MLPLayer), C.c_int32, C.c_void_p]
    L.ppenv_mlp_chain_workspace_bytes.restype = C.c_size_t
    L.ppenv_mlp_chain_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.ppenv_mlp_chain_forward.argtypes = [C.POINTER(MLPLayer), C.c_int32, C.c_void_p, C.c_void_p]
    L.ppenv_mlp_chain_status.argtypes = [C.c_void_p]
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


def layer_forward(out, x, w, bias, elu, cus=0, **kw):
    """One launch of ppenv_mlp_layer_forward on torch tensors (x: fp16 activations or fp32 observations; w: fp16 [n, k]).
    cus: size the grid for that many of the 256 CUs (ppenv_mlp_layer_forward_share; 0 = the whole chip)."""
    d = _descriptor(out, x, w, bias, elu, **kw)
    if cus:
        _lib.check(_lib_policy().ppenv_mlp_layer_forward_share(C.byref(d), cus, torch.cuda.current_stream(x.device).cuda_stream))
    else:
        _lib.check(_lib_policy().ppenv_mlp_layer_forward(C.byref(d), torch.cuda.current_stream(x.device).cuda_stream))


def chain_workspace(m, batch, count, device):
    """The zeroed workspace of ppenv_mlp_chain_forward for `count` chained layers of m rows x batch problems."""
    n = _lib_policy().ppenv_mlp_chain_workspace_bytes(m, batch, count)
    assert n > 0
    return torch.zeros((n + 3) // 4, dtype=torch.int32, device=device)


def chain_forward(descriptors, workspace):
    """ppenv_mlp_chain_forward on a list of layer descriptors (_descriptor): consecutive hidden layers in one launch."""
    arr = (MLPLayer * len(descriptors))(*descriptors)
    _lib.check(_lib_policy().ppenv_mlp_chain_forward(arr, len(descriptors), workspace.data_ptr(), torch.cuda.current_stream(workspace.device).cuda_stream))


def chain_status(workspace):
    return _lib_policy().ppenv_mlp_chain_status(workspace.data_ptr())


def heads_sample(out, x, w, bias, num_actions, actions, sigma, seed, counter, lo=-1.0, hi=1.0, neglogp=None):
    """ppenv_mlp_heads_sample: the heads layer out [M, n <= 32] (fp32) = x . w^T + bias and, in the same launch, what
    sample_actions(actions, out[:, :num_actions], sigma, seed, counter, lo, hi, neglogp) would produce."""
    d = _descriptor(out, x, w, bias, False)
    assert actions.is_contiguous() and actions.shape[1] == num_actions and sigma.is_contiguous()
    _lib.check(_lib_policy().ppenv_mlp_heads_sample(C.byref(d), num_actions, sigma.data_ptr(), seed, counter, lo, hi, actions.data_ptr(),
                                                    neglogp.data_ptr() if neglogp is not None else None,
                                                    torch.cuda.current_stream(x.device).cuda_stream))


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


def layer_backward_input(dx, dz, wt, elu_out=None, colsum_partial=None, batch=1, dz_stride=0, wt_stride=0, dx_stride=0, elu_out_stride=0,
                         colsum_stride=0, m=None, n=None, k=None):
    """ppenv_mlp_layer_backward_input on torch tensors: dx [M, n] (fp16) = (dz [M, k] . wt [n, k]^T) * ELU'(elu_out) — n = the forward
    layer's input width, k = its output width, wt = its weights transposed; colsum_partial [ceil(M / 64), >= n] fp32 receives the
    per-64-row column sums of dx (the bias gradient of the layer below, once reduce_rows has summed them)."""
    d = _descriptor(dx, dz, wt, None, False, batch=batch, in_stride=dz_stride, w_stride=wt_stride, out_stride=dx_stride, m=m, n=n, k=k)
    _lib.check(_lib_policy().ppenv_mlp_layer_backward_input(
        C.byref(d), _ptr(elu_out), elu_out_stride, elu_out.stride(0) if elu_out is not None else 0,
        _ptr(colsum_partial), colsum_stride, colsum_partial.stride(0) if colsum_partial is not None else 0, _stream(dz)))


def _dw_descriptor(dw, dz, x, batch, dz_stride, x_stride, dw_stride, m, n, k, accumulate, splits, workspace):
    d = MLPDw()
    d.m, d.n, d.k, d.batch = (dz.shape[0] if m is None else m), (dw.shape[-2] if n is None else n), (dw.shape[-1] if k is None else k), batch
    d.dz, d.dz_stride, d.lddz = dz.data_ptr(), dz_stride, dz.stride(0)
    d.x, d.x_stride, d.ldx = x.data_ptr(), x_stride, x.stride(0)
    d.dw, d.dw_stride, d.lddw = dw.data_ptr(), dw_stride, dw.stride(-2)
    d.accumulate, d.splits = int(accumulate), int(splits)
    d.workspace, d.workspace_bytes = _ptr(workspace), (workspace.numel() * workspace.element_size() if workspace is not None else 0)
    return d


def dw_workspace_bytes(m, n, k, batch=1, splits=0):
    d = MLPDw()
    d.m, d.n, d.k, d.batch, d.splits = m, n, k, batch, splits
    return int(_lib_policy().ppenv_mlp_dw_workspace_bytes(C.byref(d)))


def layer_backward_weight(dw, dz, x, batch=1, dz_stride=0, x_stride=0, dw_stride=0, m=None, n=None, k=None, accumulate=False, splits=0, workspace=None):
    """ppenv_mlp_layer_backward_weight on torch tensors: dw [n, k] fp32 (+)= dz [M, n]^T . x [M, k] (fp16 operands, fp32 accumulation on
    the matrix cores; transposed LDS reads).  workspace: a byte tensor of dw_workspace_bytes(...) (allocated here when missing)."""
    if workspace is None:
        need = dw_workspace_bytes(dz.shape[0] if m is None else m, dw.shape[-2] if n is None else n, dw.shape[-1] if k is None else k, batch, splits)
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=dz.device)
    d = _dw_descriptor(dw, dz, x, batch, dz_stride, x_stride, dw_stride, m, n, k, accumulate, splits, workspace)
    _lib.check(_lib_policy().ppenv_mlp_layer_backward_weight(C.byref(d), _stream(dz)))


def reduce_rows(out, partial, rows=None, n=None, accumulate=False):
    """out [n] (+)= the sum of partial's first `rows` rows (fp32, in row order)."""
    rows = partial.shape[0] if rows is None else rows
    n = out.numel() if n is None else n
    _lib.check(_lib_policy().ppenv_mlp_reduce_rows(partial.data_ptr(), rows, partial.stride(0), n, out.data_ptr(), int(accumulate), _stream(out)))


def bias_grad_f32(out, dz, accumulate=False, workspace=None):
    """out [n] fp32 (+)= column sums of dz [M, n] fp32 (the heads' bias gradient)."""
    L = _lib_policy()
    m, n = dz.shape
    if workspace is None:
        workspace = torch.empty(int(L.ppenv_mlp_bias_grad_workspace_bytes(m, n)), dtype=torch.uint8, device=dz.device)
    _lib.check(L.ppenv_mlp_bias_grad_f32(dz.data_ptr(), m, n, dz.stride(0), workspace.data_ptr(), out.data_ptr(), int(accumulate), _stream(dz)))


def cast_weights(w32, w16=None, wt16=None):
    """fp32 master weights [n, k] -> w16 [n, >= k] (zero-padded rows) and / or wt16 [>= k, >= n] = the transpose (zero-padded), fp16."""
    n, k = w32.shape
    _lib.check(_lib_policy().ppenv_mlp_cast_weights(w32.data_ptr(), n, k, w32.stride(0), _ptr(w16), w16.stride(0) if w16 is not None else 0,
                                                    _ptr(wt16), wt16.stride(0) if wt16 is not None else 0, wt16.shape[0] if wt16 is not None else 0, _stream(w32)))


def cast_item(w32, w16=None, wt16=None):
    """One entry of a cast_weights_batch list (the tensors must stay alive and in place: the entry holds their addresses)."""
    it = MLPCast()
    it.w32, it.n, it.k, it.ldw32 = w32.data_ptr(), w32.shape[0], w32.shape[1], w32.stride(0)
    it.w16, it.ldw16 = _ptr(w16), (w16.stride(0) if w16 is not None else 0)
    it.wt16, it.ldwt16, it.wt_rows = _ptr(wt16), (wt16.stride(0) if wt16 is not None else 0), (wt16.shape[0] if wt16 is not None else 0)
    return it


def cast_weights_batch(items, stream_of):
    """ppenv_mlp_cast_weights_batch: every matrix in `items` (cast_item entries, at most 32) in one launch."""
    arr = (MLPCast * len(items))(*items)
    _lib.check(_lib_policy().ppenv_mlp_cast_weights_batch(arr, len(items), _stream(stream_of)))


class RunningMeanStd(torch.nn.Module):
    """rl_games' RunningMeanStd (normalize_input: True, cfg/train/HumanoidPingpongTiltG1PPO.yaml:51) with the training-mode update as
    one launch (ppenv_running_mean_std_update): float64 running mean / var / count as rl_games keeps them, plus the fp32 mean and
    1 / sqrt(var + eps) the policy's normaliser reads.  rl_games itself is absent from the reference: restated from its published
    running_mean_std.py (unbiased batch variance, parallel-moments merge) — parity unpinned.
    A torch module with rl_games' own buffer names (`running_mean`, `running_var`, `count`), so that a module holding it as
    `running_mean_std` saves and restores the statistics under rl_games' checkpoint keys; the two fp32 tensors the kernels read are
    derived (non-persistent) and are refreshed in place after every load, so whoever holds them (NativeMLP.set_normalization_tensors)
    keeps seeing the current statistics."""

    def __init__(self, num_obs, device, eps=1e-5):
        super().__init__()
        self.device, self.eps, self.num_obs = torch.device(device), float(eps), int(num_obs)
        self.register_buffer("running_mean", torch.zeros(num_obs, dtype=torch.float64, device=self.device))
        self.register_buffer("running_var", torch.ones(num_obs, dtype=torch.float64, device=self.device))
        self.register_buffer("count", torch.ones((), dtype=torch.float64, device=self.device))
        self.register_buffer("mean", torch.zeros(num_obs, dtype=torch.float32, device=self.device), persistent=False)
        self.register_buffer("inv_std", torch.rsqrt(torch.ones(num_obs, dtype=torch.float32, device=self.device) + eps), persistent=False)
        self._ws, self._ws_rows = None, 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refresh())

    def refresh(self):
        """mean / inv_std (fp32, what the normaliser reads) from the float64 running statistics, IN PLACE."""
        with torch.no_grad():
            self.mean.copy_(self.running_mean.float())
            self.inv_std.copy_(torch.rsqrt(self.running_var.float() + self.eps))

    def update(self, obs):
        """obs [M, num_obs] fp32 on the device: one pass; mean / inv_std are refreshed in place (a NativeMLP given these tensors by
        set_normalization_tensors sees the new statistics at its next forward)."""
        L = _lib_policy()
        m, k = obs.shape
        assert obs.dtype == torch.float32 and obs.device == self.device and obs.stride(1) == 1 and k == self.num_obs
        if self._ws is None or self._ws_rows < m:
            self._ws = torch.zeros(int(L.ppenv_running_mean_std_workspace_bytes(m, k)) // 8 + 1, dtype=torch.float64, device=self.device)   # zeroed: the ticket
            self._ws_rows = m
        _lib.check(L.ppenv_running_mean_std_update(obs.data_ptr(), m, k, obs.stride(0), self.running_mean.data_ptr(), self.running_var.data_ptr(),
                                                   self.count.data_ptr(), self.mean.data_ptr(), self.inv_std.data_ptr(), self.eps, self._ws.data_ptr(), _stream(obs)))


class NativeMLP:
    """Actor + critic forward on the MFMA kernel.  `actor` / `critic`: lists of (weight [out, in], bias [out]) fp32 tensors, hidden
    layers first, the head last (what `[m for m in net if isinstance(m, nn.Linear)]` yields for the reference's architecture)."""

    def __init__(self, actor, critic, num_obs, device, mean=None, var=None, eps=1e-5, clip=5.0, max_rows=None, fuse_input=False, cus=0):
        """cus: the share of the chip's 256 CUs each layer launch is sized for (0 = all): 128 when two env groups' forwards run side by
        side on two streams (collector.PipelinedRollout); results do not depend on it.
        fuse_input: layer 1 reads the fp32 observations in place and normalises while staging (one launch fewer, but the
        register-staged kernel); default: a small normalise-and-pad launch first, then layer 1 on the LDS-DMA kernel like the rest
        (M = 4096, 313 observations: 54 us fused, see DESIGN.md §5a for the split path)."""
        self.device = torch.device(device)
        self.fuse_input, self.cus = bool(fuse_input), int(cus)
        # chain = (first, last): the hidden layers first .. last (1-based, first >= 2) run as ONE launch (ppenv_mlp_chain_forward); PPENV_MLP_CHAIN="3,6" sets it
        env_chain = os.environ.get("PPENV_MLP_CHAIN")
        self.chain = tuple(int(x) for x in env_chain.split(",")) if env_chain else None
        self._chain_ws = None
        assert len(actor) == len(critic) and all(a[0].shape[0] == c[0].shape[0] for a, c in zip(actor[:-1], critic[:-1]))
        self.num_obs, self.clip = int(num_obs), float(clip)
        self.units = [a[0].shape[0] for a in actor[:-1]]
        self.num_actions = actor[-1][0].shape[0]
        self.load(actor, critic)
        self.set_normalization(mean, var, eps)
        self._rows = 0
        if max_rows:
            self._alloc(max_rows)

    def sibling(self, max_rows=None):
        """Another forward context on the SAME operand images and statistics (no copy: one set of weights in HBM / L2) with its own
        activation buffers — one per env group when groups of envs are stepped on separate streams (collector.PipelinedRollout)."""
        other = object.__new__(NativeMLP)
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("h", "head_out", "mu", "value", "x16", "_rows", "_chain_ws")})
        other._rows, other._chain_ws = 0, None        # (the chain workspace belongs to one launch sequence: groups on separate streams each get their own)
        if max_rows:
            other._alloc(max_rows)
        return other

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

    def set_normalization_tensors(self, mean, inv_std):
        """Use the caller's fp32 [num_obs] statistics tensors in place (RunningMeanStd.mean / .inv_std: updated by its kernel, read by
        the next forward without a copy)."""
        assert mean.dtype == inv_std.dtype == torch.float32 and mean.device == self.device and mean.numel() == inv_std.numel() == self.num_obs
        self.mean, self.inv_std = mean, inv_std

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
        prepare_input(self.x16, env.obs_buf, self.mean, self.inv_std, self.clip)    # the rows of the CURRENT obs_buf: the first forward runs before any step has written them

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
            layer_forward(self.h[0], obs, w0, self.b[0].view(-1), elu=True, cus=self.cus, mean=self.mean, inv_std=self.inv_std, clip=self.clip, k=self.num_obs)
        else:
            if not prepared:
                prepare_input(self.x16, obs, self.mean, self.inv_std, self.clip)
            layer_forward(self.h[0], self.x16, w0, self.b[0].view(-1), elu=True, cus=self.cus)
        hidden = lambda i: dict(out=self.h[i], x=self.h[i - 1], w=self.w[i], bias=self.b[i], elu=True, batch=2, in_stride=u[i - 1], w_stride=u[i] * self.w[i].shape[-1],
                                bias_stride=u[i], out_stride=u[i], m=m, n=u[i], k=u[i - 1])
        i = 1
        while i < len(u):
            if self.chain and not self.cus and i == self.chain[0] - 1:          # layers chain[0] .. chain[1] in one launch
                last = min(self.chain[1], len(u))
                if self._chain_ws is None or self._chain_rows != m:
                    self._chain_ws, self._chain_rows = chain_workspace(m, 2, last - i, self.device), m
                chain_forward([_descriptor(**hidden(k)) for k in range(i, last)], self._chain_ws)
                i = last
                continue
            layer_forward(cus=self.cus, **hidden(i))
            i += 1
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

    def forward_net(self, which, obs, prepared=False, sample=None, head_out=None):
        """ONE of the two networks (which = 0: the actor -> mu, and with `sample` the drawn actions; 1: the critic -> value) as its own launch
        sequence of single-problem layers on the same operand images and activation buffers as forward().  For running the two networks on
        separate streams: only the actor is in front of the env step (tools/gpu_rollout_split.py; measured, not the default)."""
        m = obs.shape[0]
        assert m == self._rows and not self.fuse_input
        u, na = self.units, self.num_actions
        cols = lambda t, n: t[:, which * n:(which + 1) * n]
        if which == 0 and not prepared:
            prepare_input(self.x16, obs, self.mean, self.inv_std, self.clip)
        layer_forward(cols(self.h[0], u[0]), self.x16, self.w[0][which], self.b[0][which], elu=True, cus=self.cus)
        for i in range(1, len(u)):
            layer_forward(cols(self.h[i], u[i]), cols(self.h[i - 1], u[i - 1]), self.w[i][which], self.b[i][which], elu=True, cus=self.cus)
        ho = self.head_out if head_out is None else head_out
        ul = u[-1]
        if which == 0:
            x, w, b, out = cols(self.h[-1], ul), self.head_w[:na, :ul], self.head_b[:na], ho[:, :na]
            if sample is not None:
                heads_sample(out, x, w, b, na, sample["actions"], sample["sigma"], sample["seed"], sample["counter"], sample.get("lo", -1.0), sample.get("hi", 1.0),
                             sample.get("neglogp"))
            else:
                layer_forward(out, x, w, b, elu=False)
            return out
        out = ho[:, na:]
        layer_forward(out, cols(self.h[-1], ul), self.head_w[na:, ul:], self.head_b[na:], elu=False)
        return out

    @staticmethod
    def flops(m, num_obs, units=UNITS, num_actions=0):
        dims = [num_obs] + list(units)
        per_net = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        return 2 * m * (2 * per_net + units[-1] * (num_actions + 1))


class NativeMLPLearner:
    """The learner's forward + backward of the same network on the matrix cores (rl_games' a2c minibatch step under
    mixed_precision, cfg/train/HumanoidPingpongTiltG1PPO.yaml:50,73-76).  fp32 MASTER parameters live here in the layout of the fp16
    operand images (per hidden layer `w32[i]` [2, n, k] = actor | critic and `b32[i]` [2, n]; the heads `mu_w` [A, u], `mu_b` [A],
    `value_w` [1, u], `value_b` [1]); `sync_weights()` is the per-optimizer-step cast (both operand images in one launch per matrix);
    `forward(obs)` is NativeMLP's (its activation buffers are what the backward reads); `backward(d_head)` takes d loss / d [mu | value]
    ([M, A + 1] fp32, already multiplied by the loss scale if one is used) and fills `grads` — fp32 tensors of the parameters' shapes.
    An optimizer (rl_games: Adam) steps on `parameters()` / `grads` and calls `sync_weights()`; that part stays PyTorch."""

    def __init__(self, actor, critic, num_obs, device, mean=None, var=None, eps=1e-5, clip=5.0):
        self.net = NativeMLP(actor, critic, num_obs, device, mean=mean, var=var, eps=eps, clip=clip)
        self.device = self.net.device
        f = lambda t: t.detach().to(self.device, torch.float32).contiguous()
        self.w32 = [torch.stack([f(wa), f(wc)]).contiguous() for (wa, _), (wc, _) in zip(actor[:-1], critic[:-1])]
        self.b32 = [torch.stack([f(ba), f(bc)]).contiguous() for (_, ba), (_, bc) in zip(actor[:-1], critic[:-1])]
        self.mu_w, self.mu_b, self.value_w, self.value_b = f(actor[-1][0]), f(actor[-1][1]), f(critic[-1][0]), f(critic[-1][1])
        u, na, net = self.net.units, self.net.num_actions, self.net
        assert all(x % 64 == 0 for x in u), "hidden widths must be multiples of 64 (the LDS-DMA tile kernels' K step)"
        self.nh = (na + 1 + 7) // 8 * 8                                   # head columns padded to whole 16-byte chunks
        z16 = lambda *shape: torch.zeros(shape, dtype=torch.float16, device=self.device)
        z32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)
        # transposed operand images for dX: layer i >= 1 [2, k_i, n_i]; the heads [2 u, nh] (block-diagonal like head_w)
        self.wt = [None] + [z16(2, u[i - 1], u[i]) for i in range(1, len(u))]
        self.head_w32 = z32(na + 1, 2 * u[-1])                              # the block-diagonal master image the head launches read
        self.head_wt = z16(2 * u[-1], self.nh)
        self.grads = dict(w=[z32(*w.shape) for w in net.w], b=[z32(2, n) for n in u], head_w=z32(self.nh, 2 * u[-1]), head_b=z32(na + 1))
        self._rows = 0
        self.rms = None
        self._cast_items = None
        self.generation = 0                 # forward passes so far: backward() differentiates the LAST one (the activation buffers are shared)
        self.sync_weights()

    def parameters(self):
        return self.w32 + self.b32 + [self.mu_w, self.mu_b, self.value_w, self.value_b]

    def gradients(self):
        """Gradients in the order and shapes of parameters() (views of `grads`; layer 1's weight gradient without its K padding)."""
        g, u, na = self.grads, self.net.units, self.net.num_actions
        return ([g["w"][i][..., :self.w32[i].shape[-1]] for i in range(len(u))] + g["b"] +
                [g["head_w"][:na, :u[-1]], g["head_b"][:na], g["head_w"][na:na + 1, u[-1]:], g["head_b"][na:]])

    def sync_weights(self):
        """fp32 masters -> the fp16 operand images of the forward (w, zero-padded rows) and of dX (wt, transposed), and the fp16 biases: ONE launch for the
        whole network (ppenv_mlp_cast_weights_batch; the heads' block-diagonal master image is refreshed by two small copies first)."""
        net, u, na = self.net, self.net.units, self.net.num_actions
        if getattr(self, "_cast_items", None) is None:
            items = []
            for i in range(len(u)):
                for j in range(2):
                    items.append(cast_item(self.w32[i][j], net.w[i][j], self.wt[i][j] if i else None))
                items.append(cast_item(self.b32[i].view(1, -1), net.b[i].view(1, -1)))          # a bias vector = a one-row matrix
            items.append(cast_item(self.head_w32, net.head_w, self.head_wt))
            items.append(cast_item(self.mu_b.view(1, -1), net.head_b[:na].view(1, -1)))
            items.append(cast_item(self.value_b.view(1, -1), net.head_b[na:].view(1, -1)))
            self._cast_items = items
        self.head_w32[:na, :u[-1]] = self.mu_w
        self.head_w32[na:, u[-1]:] = self.value_w
        cast_weights_batch(self._cast_items, self.head_w32)

    def _alloc(self, m):
        u, dev = self.net.units, self.device
        wmax = 2 * max(u)
        self.dz = [torch.empty((m, wmax), dtype=torch.float16, device=dev) for _ in range(2)]      # ping-pong: dz of layer i / i - 1
        self.dhead16 = torch.zeros((m, self.nh), dtype=torch.float16, device=dev)
        self.colsum = torch.empty(((m + 63) // 64, wmax), dtype=torch.float32, device=dev)
        need = max([dw_workspace_bytes(m, 2 * u[0], self.net.w[0].shape[-1])] + [dw_workspace_bytes(m, u[i], u[i - 1], batch=2) for i in range(1, len(u))] +
                   [dw_workspace_bytes(m, self.nh, 2 * u[-1]), 16])
        self.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        self.ws_b = torch.empty(int(_lib_policy().ppenv_mlp_bias_grad_workspace_bytes(m, self.net.num_actions + 1)) + 16, dtype=torch.uint8, device=dev)
        self._rows = m

    def attach_running_mean_std(self, rms):
        """Input statistics that learn (normalize_input: True): `forward(obs, update_stats=True)` first merges the batch into `rms`
        (RunningMeanStd.update, one launch), then normalises with the refreshed statistics — rl_games' order in training mode."""
        self.rms = rms
        self.net.set_normalization_tensors(rms.mean, rms.inv_std)

    def forward(self, obs, update_stats=False):
        if update_stats:
            self.rms.update(obs)
        self.generation += 1
        return self.net.forward(obs)

    def backward(self, d_head, accumulate=False, on_grads=None, generation=None):
        """d_head [M, A + 1] fp32 = d loss / d [mu | value] of the rows of the last forward (M a multiple of 64).
        on_grads(name, tensors): called once per layer — heads first, layer 1 last — right after the launches that produce that layer's weight and
        bias gradients have been enqueued (distributed.GradientBuckets: the gradient all-reduce of a layer overlaps the backward of the layers below)."""
        net, u, na = self.net, self.net.units, self.net.num_actions
        m = d_head.shape[0]
        if generation is not None and generation != self.generation:
            raise RuntimeError(f"NativeMLPLearner.backward: the gradient belongs to forward #{generation}, but the activation buffers hold forward "
                               f"#{self.generation} — one forward, then its backward (a second forward in between overwrites what the backward reads)")
        if accumulate and on_grads is not None:
            raise ValueError("backward(accumulate=True, on_grads=...): the buffers would hold sums an earlier collective already reduced; accumulate "
                             "locally and hand on_grads the totals once (distributed.GradientBuckets)")
        assert m == net._rows and m % 64 == 0 and d_head.dtype == torch.float32 and d_head.shape[1] == na + 1 and d_head.stride(1) == 1
        if m != self._rows:
            self._alloc(m)
        g, nl, blocks = self.grads, len(u), (m + 63) // 64
        prepare_input(self.dhead16, d_head)                                # cast + pad to whole chunks (no statistics)
        bias_grad_f32(g["head_b"], d_head, accumulate, self.ws_b)
        layer_backward_weight(g["head_w"], self.dhead16, net.h[-1], accumulate=accumulate, workspace=self.ws)
        if on_grads is not None:
            on_grads("heads", [g["head_w"], g["head_b"]])
        cur = self.dz[(nl - 1) & 1][:, :2 * u[-1]]
        layer_backward_input(cur, self.dhead16, self.head_wt, elu_out=net.h[-1], colsum_partial=self.colsum)
        for i in range(nl - 1, -1, -1):
            n = u[i]
            reduce_rows(g["b"][i].view(-1), self.colsum, rows=blocks, n=2 * n, accumulate=accumulate)
            if i == 0:                                                     # both networks read the same rows: one problem over the stacked weights
                layer_backward_weight(g["w"][0].view(2 * n, -1), cur, net.x16, accumulate=accumulate, workspace=self.ws)
                if on_grads is not None:
                    on_grads("layer1", [g["w"][0], g["b"][0]])
                break
            k = u[i - 1]
            layer_backward_weight(g["w"][i], cur, net.h[i - 1], batch=2, dz_stride=n, x_stride=k, dw_stride=n * k, m=m, n=n, k=k,
                                  accumulate=accumulate, workspace=self.ws)
            if on_grads is not None:
                on_grads(f"layer{i + 1}", [g["w"][i], g["b"][i]])
            nxt = self.dz[(i - 1) & 1][:, :2 * k]
            layer_backward_input(nxt, cur, self.wt[i], elu_out=net.h[i - 1], colsum_partial=self.colsum, batch=2, dz_stride=n, wt_stride=k * n,
                                 dx_stride=k, elu_out_stride=k, colsum_stride=k, m=m, n=k, k=n)
            cur = nxt
        return self.gradients()


class _ActorCriticFn(torch.autograd.Function):
    """autograd edge of NativeActorCritic: forward = NativeMLPLearner.forward, backward = NativeMLPLearner.backward."""

    @staticmethod
    def forward(ctx, module, obs, *params):          # params: only so that autograd routes their gradients here
        mu, value = module.learner.forward(obs, update_stats=module.training and module.learner.rms is not None)
        ctx.module = module
        ctx.generation = module.learner.generation   # the activation buffers are the learner's own: backward checks nobody forwarded since
        return mu.clone(), value.clone()             # the network's own buffers are reused by the next forward

    @staticmethod
    def backward(ctx, d_mu, d_value):
        lr = ctx.module.learner
        m, na = lr.net._rows, lr.net.num_actions
        d_head = torch.zeros((m, na + 1), dtype=torch.float32, device=lr.device)
        if d_mu is not None:
            d_head[:, :na] = d_mu
        if d_value is not None:
            d_head[:, na:] = d_value
        sync = ctx.module.grad_sync                   # distributed.GradientBuckets (or None): per-layer all-reduce beside the rest of the backward
        grads = lr.backward(d_head, on_grads=sync, generation=ctx.generation)
        if sync is not None:
            sync.wait()                              # the clones below are ordered after the collectives
        return (None, None) + tuple(g.clone() for g in grads)


class NativeActorCritic(torch.nn.Module):
    """rl_games' a2c network of the reference (`separate: True` actor / critic MLPs, ELU, linear mu and value heads, fixed sigma,
    `normalize_input`; cfg/train/HumanoidPingpongTiltG1PPO.yaml:10-31,50-52) as a torch module whose forward AND backward run on the native
    kernels: `mu, value = net(obs)` are ordinary differentiable tensors, `loss.backward()` fills the fp32 `.grad` of the parameters (through
    NativeMLPLearner.backward: a GradScaler's scaled loss gives scaled gradients, `unscale_` works as on any fp32 gradient), any torch
    optimizer steps them, and the next forward recasts the fp16 operand images when a parameter has changed.  Parameters are the stacked
    masters of NativeMLPLearner (`hidden_w.i` [2, n, k] = actor | critic, `hidden_b.i`, `mu_w`, `mu_b`, `value_w`, `value_b`); `from_rlgames`
    builds one from an rl_games state dict.  The minibatch must be a multiple of 64 rows (the weight-gradient kernel's contraction tile)."""

    def __init__(self, actor, critic, num_obs, device, normalize_input=True, eps=1e-5, clip=5.0):
        super().__init__()
        self.learner = NativeMLPLearner(actor, critic, num_obs, device, eps=eps, clip=clip)
        self.learner.rms = None
        if normalize_input:
            # a SUBMODULE under rl_games' own name: state_dict() / load_state_dict() carry `running_mean_std.running_mean | running_var | count`
            # (the statistics are part of a trained policy: without them a restored network would normalise with mean 0 / var 1)
            self.running_mean_std = RunningMeanStd(num_obs, device, eps=eps)
            self.learner.attach_running_mean_std(self.running_mean_std)
        lr = self.learner
        P = torch.nn.Parameter
        self.hidden_w = torch.nn.ParameterList([P(w) for w in lr.w32])       # the same storage as the learner's masters
        self.hidden_b = torch.nn.ParameterList([P(b) for b in lr.b32])
        self.mu_w, self.mu_b, self.value_w, self.value_b = P(lr.mu_w), P(lr.mu_b), P(lr.value_w), P(lr.value_b)
        self.sigma = P(torch.zeros(lr.net.num_actions, device=lr.device), requires_grad=False)   # fixed_sigma: log-std, const 0 (yaml:21-27)
        self.grad_sync = None                        # set to a distributed.GradientBuckets for data-parallel learners (multi_gpu: True)
        self._seen = self._versions()

    @classmethod
    def from_rlgames(cls, state_dict, device, **kw):
        actor, critic = layers_from_rlgames_state_dict(state_dict)
        net = cls(actor, critic, actor[0][0].shape[1], device, normalize_input="running_mean_std.running_mean" in state_dict, **kw)
        if net.learner.rms is not None:
            rms = net.learner.rms
            with torch.no_grad():
                rms.running_mean.copy_(state_dict["running_mean_std.running_mean"])
                rms.running_var.copy_(state_dict["running_mean_std.running_var"])
                rms.count.copy_(state_dict["running_mean_std.count"])
            rms.refresh()
        if "a2c_network.sigma" in state_dict:
            net.sigma.data.copy_(state_dict["a2c_network.sigma"])
        return net

    def to_rlgames_state_dict(self):
        """The inverse of from_rlgames: rl_games' key layout for this network (what `torch.save({"model": ...})` of its a2c_continuous_logstd
        model holds — `a2c_network.{actor,critic}_mlp.<2 i>.{weight,bias}`, `a2c_network.mu.*`, `a2c_network.value.*`, `a2c_network.sigma`,
        `running_mean_std.{running_mean,running_var,count}`), so that a policy trained here plays under rl_games' player or
        RLGamesPolicy.  Detached clones on the module's device.  (`value_mean_std.*` — normalize_value — belongs to the agent, not to
        this network; rl_games is absent offline: the layout is its published one, not pinned to a file.)"""
        lr = self.learner
        actor = [(self.hidden_w[i][0], self.hidden_b[i][0]) for i in range(len(lr.net.units))] + [(self.mu_w, self.mu_b)]
        critic = [(self.hidden_w[i][1], self.hidden_b[i][1]) for i in range(len(lr.net.units))] + [(self.value_w, self.value_b)]
        return rlgames_state_dict_from_layers(actor, critic, sigma=self.sigma, rms=lr.rms)

    def _ordered(self):
        return list(self.hidden_w) + list(self.hidden_b) + [self.mu_w, self.mu_b, self.value_w, self.value_b]

    def _versions(self):
        return tuple(p._version for p in self._ordered())

    def forward(self, obs):
        if obs.shape[0] % 64:
            raise ValueError(f"NativeActorCritic: {obs.shape[0]} rows; the minibatch must be a multiple of 64")
        v = self._versions()
        if v != self._seen:                          # an optimizer stepped (or a state dict was loaded): recast the operand images
            self.learner.sync_weights()
            self._seen = v
        return _ActorCriticFn.apply(self, obs, *self._ordered())


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


def rlgames_state_dict_from_layers(actor, critic, sigma=None, rms=None):
    """The inverse of layers_from_rlgames_state_dict: (weight, bias) lists, head last -> rl_games' keys (hidden layer i sits at index 2 i of
    its nn.Sequential: Linear, activation, Linear, ...), plus `a2c_network.sigma` and `running_mean_std.*` when given.  Detached clones."""
    c = lambda t: t.detach().clone()
    sd = {}
    for which, layers, head in (("actor", actor, "mu"), ("critic", critic, "value")):
        for i, (w, b) in enumerate(layers[:-1]):
            sd[f"a2c_network.{which}_mlp.{2 * i}.weight"], sd[f"a2c_network.{which}_mlp.{2 * i}.bias"] = c(w), c(b)
        sd[f"a2c_network.{head}.weight"], sd[f"a2c_network.{head}.bias"] = c(layers[-1][0]), c(layers[-1][1])
    if sigma is not None:
        sd["a2c_network.sigma"] = c(sigma)
    if rms is not None:
        for k in ("running_mean", "running_var", "count"):
            sd[f"running_mean_std.{k}"] = c(getattr(rms, k))
    return sd


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
