"""CheckpointHandler (reference: koafusion/various/_checkpoint.py:14-62): same file-name pattern
`{model}__fold_{k}__epoch_{e:03d}.pth`, keeps the newest `num_saved` files, stores a plain state_dict.
Tensors are written as standalone contiguous CPU tensors (the live parameters are views of one flat
arena; saving the views would drag the whole arena along), so the files load in the reference too."""
import logging
import os
from pathlib import Path

import torch

logging.basicConfig()
logger = logging.getLogger("handler")
logger.setLevel(logging.DEBUG)


def portable_state_dict(model):
    sd = model.module.state_dict() if hasattr(model, "module") else model.state_dict()
    return {k: v.detach().to("cpu").contiguous().clone() for k, v in sd.items()}


def save_train_state(path, model, optimizer=None, scheduler=None, **extra):
    """Everything needed to RESUME a run (the reference's handler stores the model only): model state_dict
    (standalone CPU tensors), optimizer state in torch.optim's layout (the fused Adam exports its flat moment buffers
    per parameter, so torch.optim.Adam can continue the run and vice versa), scheduler state, and `extra` (epoch,
    fold, ...)."""
    blob = dict(model=portable_state_dict(model), extra=dict(extra))
    if optimizer is not None:
        blob["optimizer"] = optimizer.state_dict()
    if scheduler is not None:
        blob["scheduler"] = scheduler.state_dict()
    torch.save(blob, path)
    return path


def load_train_state(path, model, optimizer=None, scheduler=None):
    """Inverse of save_train_state; returns the `extra` dict.  The model must already sit on its HIP device."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    target = model.module if hasattr(model, "module") else model
    target.load_state_dict(blob["model"])
    if optimizer is not None and "optimizer" in blob:
        optimizer.load_state_dict(blob["optimizer"])
    if scheduler is not None and "scheduler" in blob:
        scheduler.load_state_dict(blob["scheduler"])
    return blob.get("extra", {})


class CheckpointHandler(object):
    def __init__(self, path_root, fname_pattern=("{model_name}__fold_{fold_idx}__epoch_{epoch_idx:>03d}.pth"),
                 num_saved=1):
        self.path_root = Path(path_root)
        self.fname_pattern = fname_pattern
        self.num_saved = num_saved
        _, ext = os.path.splitext(self.fname_pattern)
        if not self.path_root.exists():
            raise ValueError(f"Path {self.path_root} does not exist")
        self._all_ckpts = sorted(self.path_root.glob("*" + ext))
        logger.info(f"Checkpoints found: {len(self._all_ckpts)}")
        self._remove_excessive_ckpts()

    def _remove_excessive_ckpts(self):
        while len(self._all_ckpts) > self.num_saved:
            try:
                os.remove(self._all_ckpts[0])
                logger.info(f"Removed ckpt: {self._all_ckpts[0]}")
                self._all_ckpts = self._all_ckpts[1:]
            except OSError:
                logger.error(f"Cannot remove {self._all_ckpts[0]}")
                break

    def get_last_ckpt(self):
        if len(self._all_ckpts) == 0:
            logger.warning(f"No checkpoints are available in {self.path_root}")
            return None
        return self._all_ckpts[-1]

    def save_new_ckpt(self, model, model_name, fold_idx, epoch_idx):
        fname = self.fname_pattern.format(model_name=model_name, fold_idx=fold_idx, epoch_idx=epoch_idx)
        path_full = Path(self.path_root, fname)
        torch.save(portable_state_dict(model), path_full)
        self._all_ckpts.append(path_full)
        self._remove_excessive_ckpts()
