"""Flat parameter / gradient arena.

All parameters of a model live in ONE fp32 device buffer `P` (conv weights in the packed
[Cout,KH,KW,Cin] order the implicit-GEMM kernels read; the nn.Parameter keeps its reference shape
(Cout,Cin,KH,KW) as a channels_last-strided view, so `state_dict()` keys/shapes match the reference:
koafusion/models/_xrNmrMcP.py:181-182 loads them with `load_state_dict`).  Gradients are written by the
backward kernels straight into the twin buffer `G`; `p.grad` is a view of it.  One flat pair means:
  * the optimizer is ONE fused Adam launch over 389 M elements (reads p,g,m,v / writes p,m,v once),
  * the data-parallel gradient exchange is RCCL all-reduce over contiguous slices of `G` (buckets are
    just offsets; no flatten/unflatten copies),
  * BatchNorm running statistics sit in a third flat buffer so rank-0 broadcast is one collective.
"""
import torch
from torch import nn

ALIGN = 64  # elements (256 B): every parameter starts 16-B aligned for dwordx4 access


def _round_up(n, a=ALIGN):
    return (n + a - 1) // a * a


class ParamArena:
    def __init__(self, module: nn.Module):
        params = []
        seen = set()
        for name, p in module.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            if p.dtype != torch.float32:
                raise TypeError(f"{name}: arena holds fp32 parameters only, got {p.dtype}")
            params.append((name, p))
        if not params:
            raise ValueError("module has no parameters")
        self.device = params[0][1].device
        self.names = [n for n, _ in params]
        self.params = [p for _, p in params]
        self.slots = {}
        off = 0
        for name, p in params:
            n = p.numel()
            self.slots[id(p)] = (off, n)
            off += _round_up(n)
        self.numel = off
        self.P = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.G = torch.zeros(off, device=self.device, dtype=torch.float32)
        for name, p in params:
            o, n = self.slots[id(p)]
            view = self._view(self.P, o, n, p)
            view.copy_(p.data)
            p.data = view
            p._koaf_grad = self._view(self.G, o, n, p)
            p._koaf_arena = self
        # float buffers (BatchNorm running_mean / running_var)
        bufs = []
        seen = set()
        for name, b in module.named_buffers():
            if b is None or id(b) in seen or b.dtype != torch.float32:
                continue
            seen.add(id(b))
            bufs.append((name, b))
        boff = 0
        self.buf_slots = []
        for name, b in bufs:
            self.buf_slots.append((name, b, boff, b.numel()))
            boff += _round_up(b.numel())
        self.B = torch.zeros(max(boff, 1), device=self.device, dtype=torch.float32)
        for name, b, o, n in self.buf_slots:
            view = self.B[o:o + n].view(b.shape)
            view.copy_(b)
            b.data = view
        self.grad_dirty = False
        self.ready_hook = None  # set by the data-parallel wrapper: called with (offset, numel)

    @staticmethod
    def _view(flat, o, n, p):
        if p.dim() == 4:
            co, ci, kh, kw = p.shape
            return flat[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat[o:o + n].view(p.shape)

    def valid(self):
        p = self.params[0]
        lo = self.P.data_ptr()
        return p.device == self.device and lo <= p.data_ptr() < lo + self.P.numel() * 4

    def slot(self, p):
        return self.slots[id(p)]

    def active_ranges(self, params_with_grad):
        """Merge the slots of the parameters that received gradients into contiguous [lo, hi) runs."""
        spans = sorted(self.slots[id(p)] for p in params_with_grad)
        runs = []
        for o, n in spans:
            hi = o + _round_up(n)
            if runs and runs[-1][1] == o:
                runs[-1][1] = hi
            else:
                runs.append([o, hi])
        return [(a, b) for a, b in runs]


def get_arena(module: nn.Module) -> ParamArena:
    """Adopt `module`'s parameters into an arena (once per device placement)."""
    a = module.__dict__.get("_koaf_arena_obj")
    if a is not None and a.valid():
        return a
    a = ParamArena(module)
    module.__dict__["_koaf_arena_obj"] = a
    return a


def is_packed(p):
    """True if a 4-D conv weight's memory is [Cout,KH,KW,Cin] contiguous."""
    return p.dim() == 4 and p.permute(0, 2, 3, 1).is_contiguous()


def packed_weight(p):
    """Tensor whose memory is the packed [Cout,KH,KW,Cin] weight (a view when already packed)."""
    return p.detach().permute(0, 2, 3, 1).contiguous() if not is_packed(p) else p.detach()


def grad_target(p):
    """-> (buffer the backward kernel should write, accumulate?).  Buffer memory is packed for 4-D."""
    g = getattr(p, "_koaf_grad", None)
    if p.grad is None and g is not None and g.device == p.device:
        return g, False
    if p.dim() == 4:
        co, ci, kh, kw = p.shape
        t = torch.empty(co, kh, kw, ci, device=p.device, dtype=p.dtype).permute(0, 3, 1, 2)
    else:
        t = torch.empty_like(p)
    return t, p.grad is not None


def deliver_grad(p, buf, accumulate):
    if accumulate:
        p.grad.add_(buf)
    else:
        p.grad = buf
    a = getattr(p, "_koaf_arena", None)
    if a is not None:
        a.grad_dirty = True
        if a.ready_hook is not None and not accumulate:
            a.ready_hook(p)
