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

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, capturable=False):
        """capturable (as in torch.optim.Adam): learning rate and step count live in device memory (koaf_adam_hyper), so that
        step() can be captured into a HIP graph and still advance on every replay (run.GraphedTrainStep); arena parameters
        only, one shared update count (every trained parameter receives a gradient every step)."""
        if lr < 0 or eps < 0 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=bool(amsgrad)))
        self.capturable = bool(capturable)
        self._dev = {}       # capturable: (id(arena), group index) -> dict(step int32[1], lr float[1], hyper float[3], lr_host)
        self._flat = {}      # id(arena) -> dict(m, v) flat moment buffers
        self._loose = {}     # id(param) -> dict(m, v) for parameters outside any arena
        self._steps = {}     # id(param) -> number of updates it has received (torch keeps `step` per parameter)
        self._pending = {}   # id(param) -> (exp_avg, exp_avg_sq) loaded before the parameter moved into its arena

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._pending:
            self._place_pending()            # (placement is final here: the forward that produced the gradients ran)
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
                    if self.capturable:
                        # (its step count would live on the host: a captured step would replay with a frozen count)
                        raise RuntimeError("koaf Adam(capturable=True) updates arena parameters only: this parameter lives "
                                           "outside the model's arena (run one forward of the model before the first step)")
                    self._step_loose(p, lr, b1, b2, eps, wd, ams=bool(group.get("amsgrad")))
            for a, plist in by_arena.values():
                stt = self._arena_state(a)
                if self.capturable:
                    self._step_capturable(a, plist, stt, group, lr, b1, b2, eps, wd)
                    continue
                # the bias correction depends on the parameter's own update count: one fused launch per contiguous run
                # of parameters with the same count (for the reference models: every trained parameter, one count)
                by_step = {}
                for p in plist:
                    n = self._steps.get(id(p), 0) + 1
                    self._steps[id(p)] = n
                    by_step.setdefault(n, []).append(p)
                vmax = self._vmax(stt, a.P) if group.get("amsgrad") else None
                for n, ps in by_step.items():
                    for lo, hi in a.active_ranges(ps):
                        ops.adam_step(a.P[lo:hi], a.G[lo:hi], stt["m"][lo:hi], stt["v"][lo:hi], hi - lo, lr, b1, b2,
                                      eps, wd, n, self._ADAMW, vmax=vmax[lo:hi] if vmax is not None else None)
                a.epoch += 1             # the weights changed under the arena's plane images (arena.ensure_planes)
        return loss

    def _dev_state(self, a, group):
        key = (id(a), id(group))
        d = self._dev.get(key)
        if d is None:
            d = dict(step=torch.full((1,), int(getattr(self, "_resume_step", 0)), dtype=torch.int32, device=a.device),
                     lr=torch.full((1,), float(group["lr"]), device=a.device), hyper=torch.zeros(3, device=a.device),
                     lr_host=float(group["lr"]), params=set())
            self._dev[key] = d
        return d

    def sync_hyper(self):
        """copy the (scheduler-driven) learning rates to their device scalars; call outside a graph capture / before a replay"""
        for group in self.param_groups:
            for (aid, gid), d in self._dev.items():
                if gid == id(group) and d["lr_host"] != float(group["lr"]):
                    d["lr"].fill_(float(group["lr"]))
                    d["lr_host"] = float(group["lr"])

    def _step_capturable(self, a, plist, stt, group, lr, b1, b2, eps, wd):
        d = self._dev_state(a, group)
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        d["params"].update(id(p) for p in plist)
        ops.adam_hyper(d["step"], d["lr"], b1, b2, d["hyper"])
        vmax = self._vmax(stt, a.P) if group.get("amsgrad") else None
        for lo, hi in a.active_ranges(plist):
            ops.adam_step(a.P[lo:hi], a.G[lo:hi], stt["m"][lo:hi], stt["v"][lo:hi], hi - lo, lr, b1, b2, eps, wd, 1,
                          self._ADAMW, hyper=d["hyper"], vmax=vmax[lo:hi] if vmax is not None else None)
        a.epoch += 1

    def _sync_steps(self):
        """capturable: the per-parameter update counts torch's state_dict layout wants, from the device counters"""
        for d in self._dev.values():
            n = int(d["step"].item())
            for pid in d["params"]:
                self._steps[pid] = n

    @staticmethod
    def _vmax(stt, like):
        """amsgrad: the running maximum of the second moment, a flat twin of `v` (created on first use)"""
        if "vmax" not in stt:
            stt["vmax"] = torch.zeros_like(like)
        return stt["vmax"]

    def _arena_state(self, a):
        stt = self._flat.get(id(a))
        if stt is None:
            stt = dict(m=torch.zeros_like(a.P), v=torch.zeros_like(a.P))
            self._flat[id(a)] = stt
        return stt

    def _loose_state(self, p):
        stt = self._loose.get(id(p))
        if stt is None:
            stt = dict(m=torch.zeros(p.numel(), device=p.device), v=torch.zeros(p.numel(), device=p.device))
            self._loose[id(p)] = stt
        return stt

    # ---- checkpointing: torch.optim.Adam's state_dict layout, so either side resumes the other's run ----------
    def _moments(self, p, create=False, amsgrad=False):
        """(exp_avg, exp_avg_sq[, max_exp_avg_sq]) of p as tensors of p's logical shape (views of the flat buffers), or None"""
        a = getattr(p, "_koaf_arena", None)
        if a is not None and a.valid():
            if id(a) not in self._flat and not create:
                return None
            stt = self._arena_state(a)
            o, n = a.slot(p)
            out = (a._view(stt["m"], o, n, p), a._view(stt["v"], o, n, p))
            return out + (a._view(self._vmax(stt, a.P), o, n, p),) if amsgrad else out
        if id(p) not in self._loose and not create:
            return None
        stt = self._loose_state(p)
        out = (stt["m"].view(p.shape), stt["v"].view(p.shape))
        return out + (self._vmax(stt, stt["v"]).view(p.shape),) if amsgrad else out

    def _place_pending(self):
        """moments loaded by load_state_dict() go to their flat buffers once the parameters' final placement is known
        (a model adopts its arena at its first forward, which may come after the optimizer state was loaded)"""
        for g in self.param_groups:
            for p in g["params"]:
                mv = self._pending.pop(id(p), None)
                if mv is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("koaf Adam updates HIP-resident parameters only (no CPU fallback)")
                dst = self._moments(p, create=True, amsgrad=len(mv) > 2)
                for d, src in zip(dst, mv):
                    d.copy_(src.to(device=p.device, dtype=torch.float32))
        self._pending = {}

    def state_dict(self):
        if self.capturable:
            self._sync_steps()
        sd = super().state_dict()            # param_groups with index lists; `state` is kept outside self.state
        params = [(p, bool(g.get("amsgrad"))) for g in self.param_groups for p in g["params"]]
        state = {}
        for idx, (p, ams) in enumerate(params):
            n = self._steps.get(id(p), 0)
            mv = self._pending.get(id(p)) or (self._moments(p, amsgrad=ams) if n else None)   # loaded but not yet placed / live
            if mv is None:
                continue                      # never updated (no gradient so far): torch has no entry either
            state[idx] = dict(step=torch.tensor(float(n)), exp_avg=mv[0].detach().to("cpu").contiguous().clone(),
                              exp_avg_sq=mv[1].detach().to("cpu").contiguous().clone())
            if len(mv) > 2:
                state[idx]["max_exp_avg_sq"] = mv[2].detach().to("cpu").contiguous().clone()
        sd["state"] = state
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        super().load_state_dict(dict(state={}, param_groups=groups))
        params = [p for g in self.param_groups for p in g["params"]]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(params):
            raise ValueError("loaded state dict has a different number of parameters")
        self._steps, self._pending = {}, {}
        if self.capturable and state_dict["state"]:
            counts = {int(round(float(st["step"]))) for st in state_dict["state"].values()}
            if len(counts) != 1:
                raise ValueError("capturable Adam keeps one update count for all parameters; the loaded state has several")
            self._resume_step = counts.pop()         # (device counters created later start here)
            for d in self._dev.values():
                d["step"].fill_(self._resume_step)
        for stt in list(self._flat.values()) + list(self._loose.values()):
            stt["m"].zero_()
            stt["v"].zero_()
            if "vmax" in stt:
                stt["vmax"].zero_()
        for key, st in state_dict["state"].items():
            p = params[ids.index(key)] if key in ids else None
            if p is None:
                raise KeyError(f"optimizer state for unknown parameter index {key}")
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state shape {tuple(st['exp_avg'].shape)} != parameter shape {tuple(p.shape)}")
            self._pending[id(p)] = (st["exp_avg"].detach().clone(), st["exp_avg_sq"].detach().clone())
            if "max_exp_avg_sq" in st:
                self._pending[id(p)] += (st["max_exp_avg_sq"].detach().clone(),)
            self._steps[id(p)] = int(round(float(st["step"])))

    def _step_loose(self, p, lr, b1, b2, eps, wd, ams=False):
        if not p.is_cuda:
            raise RuntimeError("koaf Adam updates HIP-resident parameters only (no CPU fallback)")
        stt = self._loose_state(p)
        n = self._steps.get(id(p), 0) + 1
        self._steps[id(p)] = n
        pc = p.data.contiguous().view(-1)
        g = p.grad.contiguous().view(-1)
        ops.adam_step(pc, g, stt["m"], stt["v"], pc.numel(), lr, b1, b2, eps, wd, n, self._ADAMW,
                      vmax=self._vmax(stt, stt["v"]) if ams else None)
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

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, capturable=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad,
                         capturable=capturable)


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
