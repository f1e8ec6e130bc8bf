"""A rollout horizon collected natively (the data format on the learner's side of the path).

rl_games' a2c agent plays `horizon_length` steps per epoch and keeps every per-step tensor in an experience buffer laid out
[horizon, num_actors, ...] (cfg/train/HumanoidPingpongTiltG1PPO.yaml:73 horizon_length: 32; a2c_common.py play_steps: obses, actions,
neglogpacs, values, rewards, dones, then discount_values for returns / advantages).  With PyTorch that is a dozen small indexing
kernels per step next to the forward.  Here every producer writes STRAIGHT into its horizon-major slice:

    the env step    (ppenv_ta_step)           obs[t + 1], rewards[t], dones[t]
    the heads launch (ppenv_mlp_heads_sample) mu | value [t], actions[t], neglogp[t]
    ppenv_gae                                 advantages, returns over the finished horizon

so a horizon costs exactly the launches of `horizon` rollout steps, one bootstrap forward and one GAE launch — no copies, no host
synchronisation.  The learner (rl_games / PyTorch autograd) consumes the buffers as they are.  Works with TAEnv (the 27-DoF task of
BASELINE config 5: ppenv_ta_step takes its output pointers per call) and with PPEnv (the 7-dof tasks: ppenv_step_into; the 4-actor
variant has two actor rows per env).
"""
import ctypes as C

import torch

from . import _lib
from .policy import _lib_policy


def gae(rewards, values, dones, gamma=0.99, tau=0.95, reward_scale=1.0, advantages=None, returns=None):
    """ppenv_gae on torch tensors: rewards [H, N] fp32, values [H + 1, N] (any row stride: a column view of the heads' output),
    dones [H, N] int64 -> (advantages, returns) [H, N]."""
    h, n = rewards.shape
    assert values.shape[0] == h + 1 and values.shape[1] == n and dones.dtype == torch.int64 and rewards.is_contiguous() and dones.is_contiguous()
    adv = torch.empty_like(rewards) if advantages is None else advantages
    ret = torch.empty_like(rewards) if returns is None else returns
    _lib.check(_lib_policy().ppenv_gae(rewards.data_ptr(), values.data_ptr(), values.stride(1), values.stride(0), dones.data_ptr(), h, n, gamma, tau,
                                       reward_scale, adv.data_ptr(), ret.data_ptr(), torch.cuda.current_stream(rewards.device).cuda_stream))
    return adv, ret


class RolloutCollector:
    """`collect()` plays `horizon` steps of env + policy and leaves the horizon in `obs [H+1, N, num_obs]`, `actions [H, N, A]`,
    `neglogp [H, N]`, `mu [H+1, N, A]`, `values [H+1, N]`, `rewards [H, N]`, `dones [H, N]`, `advantages`, `returns [H, N]`."""

    def __init__(self, env, net, horizon=32, gamma=0.99, tau=0.95, reward_scale=0.01, sigma=None, seed=0):
        self.env, self.net, self.h = env, net, int(horizon)
        self.gamma, self.tau, self.reward_scale, self.seed = float(gamma), float(tau), float(reward_scale), int(seed)   # yaml:55-59: scale_value 0.01, gamma 0.99, tau 0.95
        n, dev, a = getattr(env, "num_rows", env.num_envs), env.device, net.num_actions        # actor rows: A * N for the 4-actor variant
        z = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
        self.obs = z(self.h + 1, n, env.obs_buf.shape[1])
        self.head = z(self.h + 1, n, a + 1)                     # mu | value, as the heads launch writes them
        self.mu, self.values = self.head[:, :, :a], self.head[:, :, a]
        self.actions, self.neglogp = z(self.h, n, a), z(self.h, n)
        self.rewards, self.dones = z(self.h, n), z(self.h, n, dt=torch.int64)
        self.advantages, self.returns = z(self.h, n), z(self.h, n)
        self.sigma = (torch.ones(a, device=dev) if sigma is None else sigma.to(dev, torch.float32)).contiguous()
        self.counter = 0
        self.obs[0].copy_(env.obs_buf)

    @torch.no_grad()
    def collect(self):
        """One horizon.  Row 0 of `obs` is the observation the previous horizon ended on."""
        for t in range(self.h):
            self.counter += 1
            self.net.forward(self.obs[t], head_out=self.head[t],
                             sample=dict(actions=self.actions[t], sigma=self.sigma, seed=self.seed, counter=self.counter, neglogp=self.neglogp[t]))
            self.env.step(self.actions[t], obs=self.obs[t + 1], rew=self.rewards[t], reset=self.dones[t])
        self.net.forward(self.obs[self.h], head_out=self.head[self.h])            # bootstrap value of the last observation
        gae(self.rewards, self.values, self.dones, self.gamma, self.tau, self.reward_scale, self.advantages, self.returns)
        return self

    def next_horizon(self):
        """Carry the last observation over as row 0 of the next horizon (one [N, num_obs] copy per horizon)."""
        self.obs[0].copy_(self.obs[self.h])
