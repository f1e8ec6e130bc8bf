"""isaacgymenvs.learning.amp_models: named for train.py:187-193; adversarial motion priors are not used by the HumanoidPingpong tasks."""


class ModelAMPContinuous:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("ModelAMPContinuous: AMP training is not part of the MI355X port (the HumanoidPingpong train yamls use a2c_continuous)")
