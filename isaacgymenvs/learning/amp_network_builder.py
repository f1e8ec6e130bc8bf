"""isaacgymenvs.learning.amp_network_builder: named for train.py:187-193; adversarial motion priors are not used by the HumanoidPingpong tasks."""


class AMPBuilder:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("AMPBuilder: AMP training is not part of the MI355X port (the HumanoidPingpong train yamls use a2c_continuous)")
