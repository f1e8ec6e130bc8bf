"""Tensor-API mode for the 27-DoF variant (tasks/humanoid_pingpong_3_actor_all_dof.py, "TA").

`TAState.post_physics_step` is the drop-in for the reference's post_physics_step (TA:1145-1192) on the
simulator tensors the task already wraps: reward (TA:1440-1690), masked reset (TA:965-1028) and the
313-wide observation (TA:867-904) in one launch (+ one tiny launch for the global count-flag clear).
The rigid-body step of this variant is not built; physics must come from the caller.
"""
import ctypes as C

import torch

from . import _lib, scene


class TAState:
    """Per-env buffers the reference keeps across steps (progress, sticky/count flags) plus the outputs."""

    def __init__(self, params, device="cuda:0"):
        self.L = _lib.lib()
        self.params = params
        self.device = torch.device(device)
        n = self.num_envs = params.num_envs
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)
        self.obs_buf = z((n, scene.TA_NUM_OBS), torch.float32)
        self.rew_buf = z((n,), torch.float32)
        self.reset_buf = z((n,), torch.int64)
        self.progress_buf = z((n,), torch.int64)
        self.flags = z((n,), torch.int32)
        self.episode = z((n,), torch.int32)
        self._any_reset = z((1,), torch.int32)

    def post_physics_step(self, rb_states, initial_rb_states, root_states, dof_states, dof_force, pre_ball_vx, reset_override=None):
        n = self.num_envs
        for t, numel in ((rb_states, n * 42 * 13), (initial_rb_states, n * 42 * 13), (root_states, n * 39), (dof_states, n * 54),
                         (dof_force, n * 27), (pre_ball_vx, n)):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device and t.numel() == numel
        ov = None
        if reset_override is not None:
            ov = reset_override.to(self.device, torch.float32).reshape(n, 5).contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.L.ppenv_ta_post_physics_step(
            C.byref(self.params), rb_states.data_ptr(), initial_rb_states.data_ptr(), root_states.data_ptr(), dof_states.data_ptr(),
            dof_force.data_ptr(), pre_ball_vx.data_ptr(), ov.data_ptr() if ov is not None else None, self.flags.data_ptr(),
            self.episode.data_ptr(), self.progress_buf.data_ptr(), self.obs_buf.data_ptr(), self.rew_buf.data_ptr(),
            self.reset_buf.data_ptr(), self._any_reset.data_ptr(), stream))
        if ov is not None:
            torch.cuda.current_stream(self.device).synchronize()   # keep `ov` alive until the kernel has read it
