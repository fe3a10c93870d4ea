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
A fourth buffer `W` holds the fp16 PLANE IMAGES of every convolution weight: the two pieces hi + lo of w * 2^e that the
MFMA kernel multiplies on its fp16 scheme (koaf.h: KoafGemm.fmt 1), cut once per optimizer step by one pair of launches
over the whole arena (koaf_wplanes_build: max |w| per weight, then the images) instead of by every block of every GEMM
for every k-tile; the kernels then move weight tiles global -> LDS directly.  Two arrangements per weight: F [2][Cout][K]
(forward) and D [2][Cin][taps*Cout] (the transposed weight, data gradient).  `ensure_planes()` rebuilds them when the
weights changed (own optimizer: epoch counter; anything else that writes parameters in place: tensor version counters).
"""
import ctypes

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
        self.epoch = 0          # bumped by the fused optimizer (it writes P through raw pointers)
        self._plane_stamp = None
        self._build_plane_table(module)

    # ---- weight plane images ------------------------------------------------------------------------------------
    def _build_plane_table(self, module):
        from ._lib import KoafWPlane
        rup = lambda v, a: (v + a - 1) // a * a
        ents, off, tile0, seen = [], 0, 0, set()
        for m in module.modules():
            if isinstance(m, nn.Conv2d):
                if m.groups != 1 or m.in_channels % 32 or m.out_channels % 32 or m.weight.dim() != 4:
                    continue            # (the 3-channel stem and the grouped 3x3 have their own kernels)
                R, taps, C = m.out_channels, m.kernel_size[0] * m.kernel_size[1], m.in_channels
            else:
                continue                # (linear layers stay on the bf16 scheme: their gradient operands carry no scale)
            w = m.weight
            if id(w) in seen or id(w) not in self.slots:
                continue
            seen.add(id(w))
            Kp, Rp = rup(taps * C, 32), rup(R, 32)
            f_off = off
            off += rup(2 * R * Kp, 64)
            d_off = off
            off += rup(2 * C * taps * Rp, 64)
            ents.append((w, KoafWPlane(src_off=self.slots[id(w)][0], f_off=f_off, d_off=d_off, tile0=tile0, R=R, taps=taps,
                                       C=C, Kp=Kp, Rp=Rp)))
            tile0 += ((R + 31) // 32) * taps * ((C + 31) // 32)
        self._plane_tiles = tile0
        self._plane_n = len(ents)
        if not ents:
            self.W = self._plane_tab = None
            return
        self.W = torch.zeros(off, device=self.device, dtype=torch.int16)
        self.Wamax = torch.zeros(len(ents), device=self.device, dtype=torch.float32)     # max |w| per weight
        arr = (KoafWPlane * len(ents))(*[e for _, e in ents])
        self._plane_tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        for i, (w, e) in enumerate(ents):
            w._koaf_wimg = (self.W[e.f_off:e.f_off + 2 * e.R * e.Kp], self.W[e.d_off:e.d_off + 2 * e.C * e.taps * e.Rp],
                            self.Wamax[i:i + 1])
        self._plane_params = [w for w, _ in ents]

    def _stamp(self):
        return (self.epoch, self.P._version, sum(p._version for p in self._plane_params))

    def ensure_planes(self):
        """(re)build the weight plane images on the current stream if any weight changed since the last build"""
        from . import ops
        if self.W is None or not ops.CONV_F16:
            return
        st = self._stamp()
        # (inside a HIP-graph capture the build is always recorded: a replayed step must rebuild from the weights the
        # previous replay's optimizer step left, whatever the host-side stamp of the capturing call says)
        if st == self._plane_stamp and not (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            return
        from ._lib import check, lib
        check(lib().koaf_wplanes_build(self.P.data_ptr(), self.W.data_ptr(), self.Wamax.data_ptr(), self._plane_tab.data_ptr(),
                                       self._plane_n, self._plane_tiles, torch.cuda.current_stream().cuda_stream),
              "wplanes_build")
        self._plane_stamp = st

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
    """Adopt `module`'s parameters into an arena (once per device placement); weight plane images current on return."""
    a = module.__dict__.get("_koaf_arena_obj")
    if a is None or not a.valid():
        a = ParamArena(module)
        module.__dict__["_koaf_arena_obj"] = a
    a.ensure_planes()
    return a


def weight_planes(p):
    """(F, D, amax) plane images of a convolution weight that lives in an arena (kept current by get_arena()), else None"""
    a = getattr(p, "_koaf_arena", None)
    from . import ops
    if a is None or not ops.CONV_F16 or a._plane_stamp is None or a._plane_stamp[:2] != (a.epoch, a.P._version):
        return None          # never built, or the weights moved on since (the caller falls back to the fp32 weight)
    img = getattr(p, "_koaf_wimg", None)
    return img if (img is not None and img[0].device == p.device) else None


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
