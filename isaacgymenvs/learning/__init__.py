"""isaacgymenvs.learning (train.py:96-99): the AMP agent / player / model / network builder train.py registers with rl_games.  The
HumanoidPingpong train yamls use the plain `a2c_continuous` algorithm (cfg/train/HumanoidPingpongTiltG1PPO.yaml:5,8); the AMP
classes are named here so that the registrations of train.py:187-193 (lambdas, evaluated only if an AMP config asks) import."""
from . import amp_continuous, amp_models, amp_network_builder, amp_players  # noqa: F401
