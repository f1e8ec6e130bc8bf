"""isaacgymenvs.pbt.pbt (train.py:81): population-based training is off by default (pbt.enabled: False) and out of scope here."""


def initial_pbt_check(cfg):
    raise NotImplementedError("population-based training (pbt.enabled=True, train.py:101-102) is not part of the MI355X port")


class PbtAlgoObserver:
    def __init__(self, cfg):
        initial_pbt_check(cfg)
