"""Optimizer / scheduler registries (reference: koafusion/various/_optimizers.py:4-67).

`Adam` / `AdamW` are fused: one HIP launch per contiguous run of the flat parameter arena (for the
reference models: one or two launches for all 389 M elements) instead of torch's per-tensor loops.  The
update rule is torch.optim.Adam's (coupled L2 weight decay, bias-corrected).  Parameters whose `.grad` is
None are skipped exactly like torch does (12 tensors of the cls-less aggregators never train).
"""
import torch
from torch import optim

from .. import ops
from ..arena import _round_up


class Adam(optim.Optimizer):
    _ADAMW = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not built")
        if lr < 0 or eps < 0 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = {}      # id(arena) -> dict(m, v, step)
        self._loose = {}     # id(param) -> dict(m, v, step) for parameters outside any arena

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            by_arena = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                a = getattr(p, "_koaf_arena", None)
                if a is not None and a.valid():
                    gv = p._koaf_grad
                    if p.grad.data_ptr() != gv.data_ptr():
                        gv.copy_(p.grad)       # a foreign gradient tensor: bring it into the arena
                        p.grad = gv
                    by_arena.setdefault(id(a), (a, []))[1].append(p)
                else:
                    self._step_loose(p, lr, b1, b2, eps, wd)
            for a, plist in by_arena.values():
                stt = self._flat.get(id(a))
                if stt is None:
                    stt = dict(m=torch.zeros_like(a.P), v=torch.zeros_like(a.P), step=0)
                    self._flat[id(a)] = stt
                stt["step"] += 1
                for lo, hi in a.active_ranges(plist):
                    ops.adam_step(a.P[lo:hi], a.G[lo:hi], stt["m"][lo:hi], stt["v"][lo:hi], hi - lo, lr, b1, b2, eps,
                                  wd, stt["step"], self._ADAMW)
        return loss

    def _step_loose(self, p, lr, b1, b2, eps, wd):
        if not p.is_cuda:
            raise RuntimeError("koaf Adam updates HIP-resident parameters only (no CPU fallback)")
        stt = self._loose.get(id(p))
        if stt is None:
            stt = dict(m=torch.zeros(p.numel(), device=p.device), v=torch.zeros(p.numel(), device=p.device), step=0)
            self._loose[id(p)] = stt
        stt["step"] += 1
        pc = p.data.contiguous().view(-1)
        g = p.grad.contiguous().view(-1)
        ops.adam_step(pc, g, stt["m"], stt["v"], pc.numel(), lr, b1, b2, eps, wd, stt["step"], self._ADAMW)
        if pc.data_ptr() != p.data.data_ptr():
            p.data.copy_(pc.view_as(p.data))

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        for group in self.param_groups:
            for p in group["params"]:
                a = getattr(p, "_koaf_arena", None)
                if a is not None:
                    a.grad_dirty = False


class AdamW(Adam):
    _ADAMW = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)


def warmup_static_decay_factor(epoch, epochs_warmup, epochs_static, warmup_factor=0.1, decay_factor=0.9):
    """lambda(epoch) of CustomWarmupStaticDecayLR (_optimizers.py:6-27): linear warm-up from warmup_factor
    to 1 over epochs_warmup, flat for epochs_static, then decay_factor ** (epochs past the flat part)."""
    flat_end = epochs_warmup + epochs_static
    if epoch <= epochs_warmup:
        return warmup_factor + (1. - warmup_factor) * epoch / float(epochs_warmup)
    if epoch <= flat_end:
        return 1.
    return decay_factor ** (epoch - flat_end)


def warmup_multistep_factor(epoch, epochs_warmup, mstep_milestones, warmup_factor=0.1, mstep_factor=0.1):
    """lambda(epoch) of CustomWarmupMultiStepLR (_optimizers.py:32-44)."""
    if epoch <= epochs_warmup:
        return warmup_factor + (1. - warmup_factor) * epoch / float(epochs_warmup)
    passed = sum(epoch >= epochs_warmup + ms for ms in mstep_milestones)
    return mstep_factor ** passed


def CustomWarmupStaticDecayLR(optimizer, epochs_warmup, epochs_static, epochs_decay, warmup_factor=0.1,
                              decay_factor=0.9, **kwargs):
    return optim.lr_scheduler.LambdaLR(
        optimizer=optimizer,
        lr_lambda=lambda e: warmup_static_decay_factor(e, epochs_warmup, epochs_static, warmup_factor, decay_factor))


def CustomWarmupMultiStepLR(optimizer, epochs_warmup, mstep_milestones, warmup_factor=0.1, mstep_factor=0.1,
                            **kwargs):
    return optim.lr_scheduler.LambdaLR(
        optimizer=optimizer,
        lr_lambda=lambda e: warmup_multistep_factor(e, epochs_warmup, mstep_milestones, warmup_factor, mstep_factor))


# same keys as _optimizers.py:47-52 / :54-67
dict_optimizers = {
    "SGD": optim.SGD,
    "Adam": Adam,
    "AdamW": AdamW,
    "RMSprop": optim.RMSprop,
}

dict_schedulers = {
    "LambdaLR": optim.lr_scheduler.LambdaLR,
    "MultiplicativeLR": optim.lr_scheduler.MultiplicativeLR,
    "StepLR": optim.lr_scheduler.StepLR,
    "MultiStepLR": optim.lr_scheduler.MultiStepLR,
    "ExponentialLR": optim.lr_scheduler.ExponentialLR,
    "CosineAnnealingLR": optim.lr_scheduler.CosineAnnealingLR,
    "ReduceLROnPlateau": optim.lr_scheduler.ReduceLROnPlateau,
    "CyclicLR": optim.lr_scheduler.CyclicLR,
    "OneCycleLR": optim.lr_scheduler.OneCycleLR,
    "CosineAnnealingWarmRestarts": optim.lr_scheduler.CosineAnnealingWarmRestarts,
    "CustomWarmupStaticDecayLR": CustomWarmupStaticDecayLR,
    "CustomWarmupMultiStepLR": CustomWarmupMultiStepLR,
}
