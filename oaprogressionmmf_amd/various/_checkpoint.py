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
    """Keeps the newest `num_saved` model checkpoints of a run directory (contract of the reference's handler,
    koafusion/various/_checkpoint.py:14-62: constructor arguments, the file-name pattern
    `{model_name}__fold_{fold_idx}__epoch_{epoch_idx:>03d}.pth`, `get_last_ckpt()` / `save_new_ckpt()`; a missing
    directory is a ValueError).  Written around one helper, `_trim()`, that deletes from the oldest end; the list of
    known checkpoints is rebuilt from the directory listing, so files added by another process are seen too."""

    def __init__(self, path_root, fname_pattern="{model_name}__fold_{fold_idx}__epoch_{epoch_idx:>03d}.pth", num_saved=1):
        self.path_root = Path(path_root)
        if not self.path_root.is_dir():
            raise ValueError(f"Path {self.path_root} does not exist")
        self.fname_pattern = fname_pattern
        self.num_saved = int(num_saved)
        self._suffix = Path(fname_pattern).suffix
        self._saved = []          # files written by this handler, in save order
        found = self._listing()
        logger.info("Checkpoints found: %d", len(found))
        self._trim()

    def _listing(self):
        """checkpoint files of the directory, oldest first.  Age = the order this handler saved them in (the reference
        appends to its list, so the file just written is always the newest: `epoch_1000` sorts before `epoch_999` by
        name, and a directory may hold several models / folds); files it did not write itself come before those, ordered
        by NAME exactly as the reference's `sorted(glob('*' + ext))` (:29) -- never by modification time, which a copy,
        an artifact download or a checkout does not preserve."""
        files = [p for p in self.path_root.iterdir() if p.is_file() and p.suffix == self._suffix]
        mine = {p: i for i, p in enumerate(self._saved)}
        return sorted(files, key=lambda p: (1, mine[p], p.name) if p in mine else (0, 0, p.name))

    def _trim(self):
        files = self._listing()
        for victim in files[:max(0, len(files) - self.num_saved)]:
            try:
                victim.unlink()
                logger.info("Removed ckpt: %s", victim)
            except OSError:
                logger.error("Cannot remove %s", victim)
                return

    def get_last_ckpt(self):
        files = self._listing()
        if not files:
            logger.warning("No checkpoints are available in %s", self.path_root)
            return None
        return files[-1]

    def save_new_ckpt(self, model, model_name, fold_idx, epoch_idx):
        target = self.path_root / self.fname_pattern.format(model_name=model_name, fold_idx=fold_idx, epoch_idx=epoch_idx)
        torch.save(portable_state_dict(model), target)
        if target in self._saved:
            self._saved.remove(target)
        self._saved.append(target)
        self._trim()
        self._saved = [p for p in self._saved if p.exists()]
        return target
