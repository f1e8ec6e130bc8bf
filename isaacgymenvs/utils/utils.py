"""isaacgymenvs.utils.utils (train.py:88): `set_np_formatting`, `set_seed`."""
import os
import random

import numpy as np
import torch


def set_np_formatting():
    """numpy print options for logs only (train.py:110)."""
    np.set_printoptions(edgeitems=30, infstr="inf", linewidth=4000, nanstr="nan", precision=2, suppress=False, threshold=10000, formatter=None)


def set_seed(seed, torch_deterministic=False, rank=0):
    """train.py:116: -1 picks a seed; otherwise the rank is added so that every rank of a multi-GPU run draws differently."""
    if seed == -1 and torch_deterministic:
        seed = 42 + rank
    elif seed == -1:
        seed = np.random.randint(0, 10000)
    else:
        seed = seed + rank
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    if torch_deterministic:
        os.environ["CUBLAS_WORKSPACE_CONFIG"] = ":4096:8"
        torch.backends.cudnn.benchmark = False
        torch.backends.cudnn.deterministic = True
        torch.use_deterministic_algorithms(True)
    else:
        torch.backends.cudnn.benchmark = True
        torch.backends.cudnn.deterministic = False
    return seed
