"""The rollout snippet of README.md, run as written (with a synthetic rl_games-layout checkpoint): python tools/readme_snippet_check.py on the GPU box."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, isaacgym_amd
from isaacgym_amd.policy import RLGamesPolicy
from test_policy_mlp import _rlgames_state_dict
sd = _rlgames_state_dict(torch, 313, (2048, 1536, 1024, 1024, 512, 512), 27, torch.Generator().manual_seed(0))
os.makedirs("gpurun_out", exist_ok=True)
torch.save({"model": sd}, "gpurun_out/ckpt.pth")
env = isaacgym_amd.make(task="HumanoidPingpongTiltNESSparse27DOFG1", num_envs=4096, sim_device="cuda:0", rl_device="cuda:0")
policy = RLGamesPolicy.load("gpurun_out/ckpt.pth", "cuda:0")
obs = env.reset()["obs"]
tot = 0.0
for _ in range(100):
    actions, values = policy.act(env.obs_buf, deterministic=True)
    obs, rew, done, info = env.step(actions)
    tot += float(rew.mean())
print("ok", obs["obs"].shape, float(values.mean()), tot, int(done.sum()))
