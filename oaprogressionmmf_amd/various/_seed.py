"""Seeding of every random source the train step draws from.

Same contract as the reference's `set_ultimate_seed` (koafusion/various/_seed.py:1-20): the hash seed and Python's and
NumPy's generators get `base_seed`, torch's CPU generator `base_seed + 1`, the device generators `base_seed + 2`, and
the cuDNN / MIOpen autotune flags are pinned.  The libkoaf dropout kernels take their per-call seeds from torch's CPU
generator (functional._seed), so they are covered by the `+ 1` stream; nothing on this path uses MIOpen -- the two
backend flags are set only so that scripts reading them see what the reference would have set."""
import importlib
import os
import random

_OFFSET_TORCH_CPU, _OFFSET_TORCH_DEVICE = 1, 2


def _optional(module_name):
    try:
        return importlib.import_module(module_name)
    except ModuleNotFoundError:
        print(f"Module `{module_name}` has not been found")
        return None


def set_ultimate_seed(base_seed=777):
    os.environ["PYTHONHASHSEED"] = str(base_seed)
    random.seed(base_seed)
    np = _optional("numpy")
    if np is not None:
        np.random.seed(base_seed)
    torch = _optional("torch")
    if torch is None:
        return
    torch.manual_seed(base_seed + _OFFSET_TORCH_CPU)
    torch.cuda.manual_seed_all(base_seed + _OFFSET_TORCH_DEVICE)
    for flag, value in (("deterministic", True), ("benchmark", False)):
        setattr(torch.backends.cudnn, flag, value)
