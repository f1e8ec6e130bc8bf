"""isaacgymenvs.learning.amp_players: named for train.py:187-193; adversarial motion priors are not used by the HumanoidPingpong tasks."""


class AMPPlayerContinuous:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("AMPPlayerContinuous: AMP training is not part of the MI355X port (the HumanoidPingpong train yamls use a2c_continuous)")
