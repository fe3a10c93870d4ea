"""set_ultimate_seed (reference: koafusion/various/_seed.py:1-20): 777 / 778 / 779."""
import os
import random


def set_ultimate_seed(base_seed=777):
    os.environ["PYTHONHASHSEED"] = str(base_seed)
    random.seed(base_seed)
    try:
        import numpy as np
        np.random.seed(base_seed)
    except ModuleNotFoundError:
        print("Module `numpy` has not been found")
    try:
        import torch
        torch.manual_seed(base_seed + 1)
        torch.cuda.manual_seed_all(base_seed + 2)
        # MIOpen is not on this path; the flags are kept so scripts that read them behave the same
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    except ModuleNotFoundError:
        print("Module `torch` has not been found")
