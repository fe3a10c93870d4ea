"""FeaT / Transformer / Attention / FeedForward with the reference's parameter names
(koafusion/models/_core_trf.py:74-205), executed through the libkoaf kernels.

Quirks kept on purpose: attention scale = dim ** -0.5 with dim = the FULL width (:160);
qkv split order '(qkv h d)' (:170); to_qkv has no bias, to_out has; pre-LN blocks without a final LN;
learned pos_embedding only; the mask path of the reference is dead code (:2,173) so mask must be None.
"""
import torch
from torch import nn

from .. import functional as KF
from .. import ops
from ..arena import deliver_grad, grad_target


class _EmbedFn(torch.autograd.Function):
    """cat(cls_token, x) + pos_embedding  (_core_trf.py:121-125); gradients of the two parameters are
    batch column-sums written into the gradient arena."""

    @staticmethod
    def forward(ctx, x, cls_token, pos):
        B = x.shape[0]
        ncls = 0
        if cls_token is not None:
            ncls = cls_token.shape[1]
            x = torch.cat((cls_token.detach().expand(B, -1, -1), x), dim=1)
        if pos.shape[1] != x.shape[1]:
            raise RuntimeError(f"pos_embedding holds {pos.shape[1]} tokens, input has {x.shape[1]}")
        out = ops.add(x.contiguous(), pos.detach().expand(B, -1, -1).contiguous())
        ctx.cls_token, ctx.pos, ctx.ncls = cls_token, pos, ncls
        return out

    @staticmethod
    def backward(ctx, dy):
        B, n, D = dy.shape
        dy = dy.contiguous()
        pos, cls_token, ncls = ctx.pos, ctx.cls_token, ctx.ncls
        from .._lib import lib, check
        L = lib()
        if pos.requires_grad:
            g, acc = grad_target(pos)
            ws = torch.empty(max(L.koaf_colsum_ws(B, n * D), 1), device=dy.device)
            check(L.koaf_colsum(dy.data_ptr(), g.data_ptr(), B, n * D, ws.data_ptr(),
                                torch.cuda.current_stream().cuda_stream), "colsum(pos)")
            deliver_grad(pos, g, acc)
        if cls_token is not None and cls_token.requires_grad:
            g, acc = grad_target(cls_token)
            dc = dy[:, :ncls].contiguous()
            ws = torch.empty(max(L.koaf_colsum_ws(B, ncls * D), 1), device=dy.device)
            check(L.koaf_colsum(dc.data_ptr(), g.data_ptr(), B, ncls * D, ws.data_ptr(),
                                torch.cuda.current_stream().cuda_stream), "colsum(cls)")
            deliver_grad(cls_token, g, acc)
        return dy[:, ncls:], None, None


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(dim, hidden_dim),
            nn.GELU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, dim),
            nn.Dropout(dropout),
        )

    def forward(self, x, residual=None):
        l0, l3 = self.net[0], self.net[3]
        p = self.net[2].p
        h = KF.gelu(KF.linear(x, l0.weight, l0.bias))
        h = KF.dropout(h, p, self.training)
        if residual is not None and not (self.training and p > 0):
            return KF.linear(h, l3.weight, l3.bias, residual=residual)
        out = KF.dropout(KF.linear(h, l3.weight, l3.bias), p, self.training)
        return out if residual is None else KF.AddFn.apply(out, residual)


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dropout=0.):
        super().__init__()
        self.heads = heads
        self.scale = dim ** -0.5
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Dropout(dropout))

    def forward(self, x, mask=None, residual=None):
        if mask is not None:
            raise NotImplementedError("the reference's mask path is dead code (_core_trf.py:2,173 raises); mask must be None")
        qkv = KF.linear(x, self.to_qkv.weight, None)
        out, attn = KF.attention(qkv, self.heads, self.scale)
        lo, p = self.to_out[0], self.to_out[1].p
        if residual is not None and not (self.training and p > 0):
            return KF.linear(out, lo.weight, lo.bias, residual=residual), attn
        out = KF.dropout(KF.linear(out, lo.weight, lo.bias), p, self.training)
        if residual is not None:
            out = KF.AddFn.apply(out, residual)
        return out, attn


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, mlp_dim, dropout):
        super().__init__()
        self.depth = depth
        for d in range(depth):
            setattr(self, f"prenorm_0_{d}", nn.LayerNorm(dim))
            setattr(self, f"attn_{d}", Attention(dim, heads=heads, dropout=dropout))
            setattr(self, f"prenorm_1_{d}", nn.LayerNorm(dim))
            setattr(self, f"ff_{d}", FeedForward(dim, mlp_dim, dropout=dropout))

    def forward(self, x, mask=None):
        attentions = []
        for d in range(self.depth):
            n0 = getattr(self, f"prenorm_0_{d}")
            o = KF.layer_norm(x, n0.weight, n0.bias, n0.eps)
            x, attn = getattr(self, f"attn_{d}")(o, mask, residual=x)   # x = attn(o) + x
            attentions.append(attn)
            n1 = getattr(self, f"prenorm_1_{d}")
            ff = KF.layer_norm(x, n1.weight, n1.bias, n1.eps)
            x = getattr(self, f"ff_{d}")(ff, residual=x)                # x = ff(.) + x
        return x, attentions


class FeaT(nn.Module):
    def __init__(self, num_patches, patch_dim, emb_dim, depth, heads, mlp_dim, num_classes, emb_dropout=0.,
                 with_cls=True, num_cls_tokens=1, mlp_dropout=0., num_outputs=1):
        super().__init__()
        self.patch_dim = patch_dim
        self.num_outputs = num_outputs
        self.with_cls = with_cls
        if self.with_cls:
            self.cls_token = nn.Parameter(torch.randn(1, num_cls_tokens, emb_dim))
        else:
            num_cls_tokens = 0
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + num_cls_tokens, emb_dim))
        self.patch_to_embedding = nn.Linear(self.patch_dim, emb_dim)
        self.emb_dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(emb_dim, depth, heads, mlp_dim, mlp_dropout)
        self.to_cls_token = nn.Identity()
        for i in range(self.num_outputs):
            setattr(self, f"mlp_head{i}", nn.Sequential(
                nn.LayerNorm(emb_dim),
                nn.Linear(emb_dim, mlp_dim),
                nn.GELU(),
                nn.Dropout(mlp_dropout),
                nn.Linear(mlp_dim, num_classes),
            ))

    def _head(self, head, x):
        ln, l1, drop, l2 = head[0], head[1], head[3], head[4]
        t = KF.layer_norm(x, ln.weight, ln.bias, ln.eps)
        t = KF.gelu(KF.linear(t, l1.weight, l1.bias))
        t = KF.dropout(t, drop.p, self.training)
        return KF.linear(t, l2.weight, l2.bias)

    def forward(self, features, mask=None, need_head=True):
        pe = self.patch_to_embedding
        x = KF.linear(features, pe.weight, pe.bias)
        x = _EmbedFn.apply(x, self.cls_token if self.with_cls else None, self.pos_embedding)
        x = KF.dropout(x, self.emb_dropout.p, self.training)
        states, attentions = self.transformer(x, mask)
        outputs = []
        if need_head:
            xs = states[:, 0:self.num_outputs]
            for i in range(self.num_outputs):
                outputs.append(self._head(getattr(self, f"mlp_head{i}"), xs[:, i]))
            if len(outputs) > 0:
                outputs = torch.stack(outputs, dim=1)
        return outputs, states, attentions
