"""isaacgymenvs.utils.wandb_utils (train.py:92): W&B logging is off by default (wandb_activate: False) and out of scope here."""


class WandbAlgoObserver:
    def __init__(self, cfg):
        raise NotImplementedError("Weights & Biases logging (wandb_activate=True, train.py:178-183) is not part of the MI355X port")
