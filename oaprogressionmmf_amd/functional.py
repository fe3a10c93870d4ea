"""torch.autograd.Function wrappers around the libkoaf kernels for the transformer-side ops.

Each Function runs hand-written HIP kernels in forward AND backward; parameter gradients are written by
the kernels directly into the model's flat gradient arena (see arena.py) and attached as `p.grad`
views, so nothing is copied and the fused Adam / RCCL all-reduce see one contiguous buffer.
"""
import torch

from . import ops
from .arena import deliver_grad, grad_target


class DeviceStepState(object):
    """Step-to-step randomness without host involvement, for train steps captured into a HIP graph (run.GraphedTrainStep).

    Eager steps draw one dropout seed per call site from torch's CPU generator; a captured step replays with frozen kernel
    arguments, so instead every call site gets a fixed salt (site index within the step, mixed with `base_seed`) and the
    kernels fold a device-resident step counter (`epoch`, bumped by one kernel at the start of every step) into it.
    While an instance is installed as `functional.STEP_STATE` eager steps use the same scheme, so eager and replayed
    steps draw identical masks."""

    def __init__(self, device, base_seed):
        self.epoch = torch.zeros(1, dtype=torch.int64, device=device)
        self.base = int(base_seed) & (2 ** 62 - 1)
        self.site = 0

    def begin_step(self):
        from . import ops as _ops
        self.site = 0
        _ops.counter_add(self.epoch, 1)

    def salt(self):
        z = (self.base + 0x9E3779B97F4A7C15 * (self.site + 1)) & (2 ** 64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        self.site += 1
        return (z ^ (z >> 31)) & (2 ** 62 - 1)


STEP_STATE = None      # a DeviceStepState while a graph-capturable train step runs (run.GraphedTrainStep installs it)


def _seed():
    """-> (seed, epoch tensor or None)"""
    if STEP_STATE is not None:
        return STEP_STATE.salt(), STEP_STATE.epoch
    # drawn from torch's CPU generator so `set_ultimate_seed` (various/_seed.py) makes runs repeatable
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()), None


class LinearFn(torch.autograd.Function):
    """nn.Linear (koafusion/models/_core_trf.py:104,144-149,161-164) with optional fused residual add."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        r2 = None
        if residual is not None:
            r2 = residual.reshape(M, N)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        y = ops.linear_fwd(x2, weight.detach(), bias.detach() if bias is not None else None, M, N, K, residual=r2)
        ctx.x2 = x2
        ctx.weight, ctx.bias = weight, bias
        ctx.has_res = residual is not None
        ctx.dims = (M, N, K)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        M, N, K = ctx.dims
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        w, b = ctx.weight, ctx.bias
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(dy2, w.detach(), M, N, K).view(ctx.xshape)
        if w.requires_grad:
            gw, accw = grad_target(w)
            gb, accb = (None, False)
            if b is not None and b.requires_grad:
                gb, accb = grad_target(b)
            ops.linear_wgrad(dy2, ctx.x2, gw, gb, M, N, K)
            deliver_grad(w, gw, accw)
            if gb is not None:
                deliver_grad(b, gb, accb)
        ctx.x2 = None
        return dx, None, None, (dy if ctx.has_res else None)


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim (koafusion/models/_core_trf.py:110,190,192)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        y, mean, rstd = ops.layernorm_fwd(x2, weight.detach(), bias.detach(), rows, D, eps)
        ctx.saved = (x2, mean, rstd)
        ctx.weight, ctx.bias = weight, bias
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved
        rows, D = x2.shape
        dy2 = dy.reshape(rows, D)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        w, b = ctx.weight, ctx.bias
        gw, accw = grad_target(w)
        gb, accb = grad_target(b)
        dx = ops.layernorm_bwd(dy2, x2, w.detach(), mean, rstd, gw, gb, rows, D)
        deliver_grad(w, gw, accw)
        deliver_grad(b, gb, accb)
        ctx.saved = None
        return dx.view(ctx.xshape), None, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(scale * Q K^T) V on the fused qkv projection with the reference's '(qkv h d)' split
    (koafusion/models/_core_trf.py:170-180).  Returns (out, attn)."""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        B, n, three_dim = qkv.shape
        d = three_dim // (3 * heads)
        q = qkv if qkv.is_contiguous() else qkv.contiguous()
        out, attn = ops.attention_fwd(q, B, n, heads, d, scale)
        ctx.saved = (q, attn)
        ctx.dims = (B, n, heads, d, scale)
        ctx.mark_non_differentiable(attn)
        return out, attn

    @staticmethod
    def backward(ctx, dout, _dattn):
        q, attn = ctx.saved
        B, n, h, d, scale = ctx.dims
        do = dout if dout.is_contiguous() else dout.contiguous()
        dqkv = ops.attention_bwd(do, q, attn, B, n, h, d, scale)
        ctx.saved = None
        return dqkv, None, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x if x.is_contiguous() else x.contiguous()
        ctx.x = xc
        return ops.gelu_fwd(xc)

    @staticmethod
    def backward(ctx, dy):
        d = dy if dy.is_contiguous() else dy.contiguous()
        dx = ops.gelu_bwd(d, ctx.x)
        ctx.x = None
        return dx


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x if x.is_contiguous() else x.contiguous()
        y = ops.relu_fwd(xc)
        ctx.y = y
        return y

    @staticmethod
    def backward(ctx, dy):
        d = dy if dy.is_contiguous() else dy.contiguous()
        dx = ops.relu_bwd(d, ctx.y)
        ctx.y = None
        return dx


class DropoutFn(torch.autograd.Function):
    """Inverted dropout; the mask is regenerated from (seed, index) in backward, never stored."""

    @staticmethod
    def forward(ctx, x, p, seed):
        xc = x if x.is_contiguous() else x.contiguous()
        ctx.p, ctx.seed = p, seed
        return ops.dropout(xc, p, seed[0], seed[1])

    @staticmethod
    def backward(ctx, dy):
        d = dy if dy.is_contiguous() else dy.contiguous()
        return ops.dropout(d, ctx.p, ctx.seed[0], ctx.seed[1]), None, None


class Dropout2dFn(torch.autograd.Function):
    """channel dropout on an (N, C, h, w) view of an NHWC buffer (the encoder output layout)"""

    @staticmethod
    def forward(ctx, x, p, seed):
        N, C, h, w = x.shape
        xc = x.permute(0, 2, 3, 1)
        xc = xc if xc.is_contiguous() else xc.contiguous()
        ctx.meta = (N, h * w, C, p, seed)
        return ops.dropout2d(xc, N, h * w, C, p, seed[0], seed[1]).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        N, HW, C, p, seed = ctx.meta
        d = dy.permute(0, 2, 3, 1)
        d = d if d.is_contiguous() else d.contiguous()
        return ops.dropout2d(d, N, HW, C, p, seed[0], seed[1]).permute(0, 3, 1, 2), None, None


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.contiguous(), b.expand_as(a).contiguous())

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class LossFn(torch.autograd.Function):
    """FocalLoss / CrossEntropyLoss forward + backward in one kernel
    (koafusion/various/_losses.py:89-108; the softmax-CE backward of BASELINE.json)."""

    @staticmethod
    def forward(ctx, logits, target, gamma, mean, focal, class_weight=None):
        lg = logits if logits.is_contiguous() else logits.contiguous()
        loss, dl = ops.focal_loss(lg, target.contiguous(), gamma, mean=mean, focal=focal, class_weight=class_weight)
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, g):
        dl = ctx.dl * g
        return dl, None, None, None, None, None


def linear(x, weight, bias=None, residual=None):
    return LinearFn.apply(x, weight, bias, residual)


def layer_norm(x, weight, bias, eps=1e-5):
    return LayerNormFn.apply(x, weight, bias, eps)


def attention(qkv, heads, scale):
    return AttentionFn.apply(qkv, heads, scale)


def gelu(x):
    return GeluFn.apply(x)


def relu(x):
    return ReluFn.apply(x)


def dropout(x, p, training):
    if not training or p == 0.0:
        return x
    return DropoutFn.apply(x, float(p), _seed())


def dropout2d(x, p, training):
    """nn.Dropout2d on an (N, C, h, w) encoder output"""
    if not training or p == 0.0:
        return x
    return Dropout2dFn.apply(x, float(p), _seed())
