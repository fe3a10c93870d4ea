"""Kernel-level parity: every libkoaf entry point against a plain torch fp32/fp64 CPU reference of the
same op (tolerances written at each assert).  All calls go through the C ABI (ctypes).

Contractions: every product -- forward, data- and weight-gradient -- is formed at fp32 rounding level (bars 2e-6 forward,
BWD = 4e-6 on the longer gradient sums; BASELINE's bar is 1e-3), on either scheme of koaf.h's KoafGemm.fmt: three bf16
pieces / six products (any operand) and two scaled fp16 pieces / three products (convolutions whose operand magnitudes are
known: weight plane images + the amax the BatchNorm backward leaves).  test_conv2d runs both; test_fp16_scheme_* check the
scale handling (tiny / huge / wide-range operands) and that the pre-split weight images give the same bits as the in-kernel
split."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import procedural as P
from common import load

from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent

pytestmark = pytest.mark.gpu

BWD = 4e-6      # gradient contractions (see the module docstring)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def nhwc(x):  # NCHW cpu -> NHWC contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def packw(w):  # (Cout,Cin,KH,KW) -> packed (Cout,KH,KW,Cin)
    return w.permute(0, 2, 3, 1).contiguous()


G = torch.Generator().manual_seed(1234)


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=G) * scale


@pytest.mark.parametrize("M,N,K", [(512, 2048, 2048), (736, 6144, 2048), (8, 2048, 2048), (8, 2, 2048), (8, 2048, 9),
                                   (200, 64, 100), (129, 130, 36)])
def test_linear(dev, M, N, K):
    from oaprogressionmmf_amd import ops
    x, w, b, r = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N), rnd(M, N)
    y_ref = (x.double() @ w.double().t() + b.double() + r.double())
    y = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), M, N, K, residual=r.to(dev))
    assert rel_err(y, y_ref) < 2e-6
    dy = rnd(M, N)
    dx_ref = dy.double() @ w.double()
    dx = ops.linear_dgrad(dy.to(dev), w.to(dev), M, N, K)
    assert rel_err(dx, dx_ref) < BWD
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    ops.linear_wgrad(dy.to(dev), x.to(dev), dw, db, M, N, K)
    assert rel_err(dw, dy.double().t() @ x.double()) < BWD
    assert rel_err(db, dy.double().sum(0)) < 2e-6


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (3, 20, 20, 64, 64, 1, 1, 0),
    (3, 20, 20, 64, 256, 1, 1, 0),
    (2, 20, 20, 256, 512, 1, 2, 0),
    (3, 20, 20, 64, 64, 3, 1, 1),
    (2, 21, 19, 128, 128, 3, 2, 1),
    (5, 10, 10, 256, 256, 3, 1, 1),
    (4, 5, 5, 512, 512, 3, 1, 1),
]


def amax_of(t):
    """device scalar max |t| (what koaf_bn_bwd_apply leaves for the gradients it writes)"""
    return t.abs().max().reshape(1).float()


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("prologue", [False, True])
@pytest.mark.parametrize("scheme", ["bf16", "f16"])
def test_conv2d(dev, case, prologue, scheme):
    from oaprogressionmmf_amd import ops
    N, H, W, Cin, Cout, k, s, p = case
    x = rnd(N, Cin, H, W)
    w = rnd(Cout, Cin, k, k, scale=(Cin * k * k) ** -0.5)
    sc, sh = (rnd(Cin) * 0.5 + 1.0), rnd(Cin) * 0.3
    xin = x.double()
    if prologue:
        xin = torch.relu(xin * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    xin.requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(xin, wd, stride=s, padding=p)
    dy = rnd(*y_ref.shape)
    y_ref.backward(dy.double())
    xd, wp = nhwc(x).to(dev), packw(w).to(dev)
    scd, shd = (sc.to(dev), sh.to(dev)) if prologue else (None, None)
    wimg = ops.build_weight_planes(wp, Cout, k * k, Cin) if scheme == "f16" else None
    y, part = ops.conv2d_fwd(xd, wp, N, H, W, Cin, Cout, k, k, s, p, scd, shd, stats=True, wimg=wimg)
    assert rel_err(nchw(y.cpu()), y_ref) < 2e-6
    # epilogue statistics = column sums / sums of squares of y
    s1 = part[:, 0].double().sum(0).cpu()
    s2 = part[:, 1].double().sum(0).cpu()
    yr = y_ref.detach().permute(1, 0, 2, 3).reshape(Cout, -1)
    assert rel_err(s1, yr.sum(1)) < 1e-4
    assert rel_err(s2, (yr * yr).sum(1)) < 1e-5
    dyd = nhwc(dy).to(dev)
    am = amax_of(dyd) if scheme == "f16" else None
    if not prologue:
        res = rnd(N, H, W, Cin)
        dx = ops.conv2d_dgrad(dyd, wp, N, H, W, Cin, Cout, k, k, s, p, residual=res.to(dev), wimg=wimg, dy_amax=am)
        assert rel_err(nchw(dx.cpu()) - nchw(res), xin.grad) < BWD
    dw = torch.empty(Cout, k, k, Cin, device=dev)
    ops.conv2d_wgrad(dyd, xd, dw, N, H, W, Cin, Cout, k, k, s, p, scd, shd, dy_amax=am)
    assert rel_err(dw.cpu().permute(0, 3, 1, 2), wd.grad) < BWD


def test_conv2d_wgrad_large_splitk(dev):
    from oaprogressionmmf_amd import ops
    N, H, W, Cin, Cout = 64, 40, 40, 64, 64
    x, w = rnd(N, Cin, H, W), rnd(Cout, Cin, 1, 1)
    dy = rnd(N, Cout, H, W)
    ref = torch.einsum("nohw,nchw->oc", dy.double(), x.double())
    dw = torch.empty(Cout, 1, 1, Cin, device=dev)
    ops.conv2d_wgrad(nhwc(dy).to(dev), nhwc(x).to(dev), dw, N, H, W, Cin, Cout, 1, 1, 1, 0)
    assert rel_err(dw.cpu().reshape(Cout, Cin), ref) < BWD


@pytest.mark.parametrize("scheme", ["bf16", "f16"])
@pytest.mark.parametrize("C,groups,stride,H", [(128, 32, 1, 22), (256, 32, 2, 22), (512, 32, 1, 11), (1024, 32, 2, 11)])
def test_gconv3x3(dev, C, groups, stride, H, scheme):
    """scheme f16: max |w| rides on the expanded weight and max |dy| on the gradient, the three calls run the fp16 scheme (three
    MFMAs per product); bf16: the scalars are withheld, six.  Same bars."""
    from oaprogressionmmf_amd import ops
    N, W = 2, H
    Cg = C // groups
    x = rnd(N, C, H, W)
    w = rnd(C, Cg, 3, 3, scale=(Cg * 9) ** -0.5)
    sc, sh = (rnd(C) * 0.5 + 1.0), rnd(C) * 0.3
    xin = torch.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    xin.requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(xin, wd, stride=stride, padding=1, groups=groups)
    dy = rnd(*y_ref.shape)
    y_ref.backward(dy.double())
    xd, wp = nhwc(x).to(dev), packw(w).to(dev)
    wexp = ops.gconv_expand_w(wp, C, groups)
    assert float(wexp._koaf_amax) == float(wp.abs().max())
    if scheme == "bf16":
        del wexp._koaf_amax
    y, part = ops.gconv3x3_fwd(xd, wexp, N, H, W, C, stride, sc.to(dev), sh.to(dev), stats=True)
    assert rel_err(nchw(y.cpu()), y_ref) < 2e-6
    yr = y_ref.detach().permute(1, 0, 2, 3).reshape(C, -1)
    assert rel_err(part[:, 0].double().sum(0).cpu(), yr.sum(1)) < 1e-4
    dyd = nhwc(dy).to(dev)
    if scheme == "f16":
        dyd._koaf_amax = dyd.abs().max().reshape(1)
    dx = ops.gconv3x3_dgrad(dyd, wexp, N, H, W, C, stride)
    assert rel_err(nchw(dx.cpu()), xin.grad) < BWD
    dwexp = ops.gconv3x3_wgrad(dyd, xd, N, H, W, C, stride, sc.to(dev), sh.to(dev))
    dw = torch.empty(C, 3, 3, Cg, device=dev)
    ops.gconv_compress_dw(dwexp, dw, C, groups)
    assert rel_err(dw.cpu().permute(0, 3, 1, 2), wd.grad) < BWD


@pytest.mark.parametrize("N,H,W", [(3, 40, 40), (2, 35, 31), (1, 70, 70), (2, 30, 401), (1, 9, 790)])     # (the last two: several column bands)
def test_stem(dev, N, H, W):
    from oaprogressionmmf_amd import ops
    x = rnd(N, 1, H, W)
    w = rnd(64, 3, 7, 7, scale=147 ** -0.5)
    x3 = x.double().repeat(1, 3, 1, 1)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(x3, wd, stride=2, padding=3)
    dy = rnd(*y_ref.shape)
    y_ref.backward(dy.double())
    wp = packw(w).to(dev)
    w1t = ops.stem_fold_w(wp)
    xd = x.reshape(N, H, W).contiguous().to(dev)
    y = ops.stem_fwd(xd, w1t, N, H, W)
    assert rel_err(nchw(y.cpu()), y_ref) < 2e-6
    # the statistics epilogue: same output, partial sums of (y - shift) and (y - shift)^2 per block
    shift = (rnd(64) * 0.1).to(dev)
    for dt in (torch.float32, torch.bfloat16):
        y2, part = ops.stem_fwd(xd, w1t, N, H, W, dtype=dt, stats=True, shift=shift)
        assert torch.equal(y2, y if dt == torch.float32 else y.bfloat16())
        d = y2.double().reshape(-1, 64) - shift.double()
        assert rel_err(part[:, 0].double().sum(0), d.sum(0)) < 1e-5 and rel_err(part[:, 1].double().sum(0), (d * d).sum(0)) < 1e-5
    dw = torch.empty(64, 7, 7, 3, device=dev)
    ops.stem_wgrad(nhwc(dy).to(dev), xd, dw, N, H, W)
    # folded gradient: every one of the 3 input channels sees the same image, so dW[:,c] are identical
    assert rel_err(dw.cpu().permute(0, 3, 1, 2), wd.grad) < BWD


@pytest.mark.parametrize("rows,C", [(5000, 64), (777, 256), (300, 2048)])
def test_batchnorm_train(dev, rows, C):
    from oaprogressionmmf_amd import ops
    c = rnd(rows, C) * 2.0 + rnd(C)[None, :]
    gamma, beta = rnd(C) * 0.5 + 1.0, rnd(C) * 0.2
    rm, rv = rnd(C) * 0.1, torch.rand(C, generator=G) + 0.5
    cd = c.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    z = F.batch_norm(cd, rm_ref, rv_ref, gd, bd, training=True, momentum=0.1, eps=1e-5)
    a = torch.relu(z)
    g = rnd(rows, C)
    a.backward(g.double())
    cdv = c.to(dev)
    part = ops.colstats(cdv, rows, C)
    rmd, rvd = rm.to(dev), rv.to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    saved = ops.bn_finalize(part, C, rows, gamma.to(dev), beta.to(dev), rmd, rvd, nbt, 0.1, 1e-5, True)
    assert nbt.item() == 1
    assert rel_err(rmd, rm_ref) < 1e-6 and rel_err(rvd, rv_ref) < 1e-6
    a_k = ops.bn_add_relu(cdv, saved, rows, C)
    assert rel_err(a_k, a) < 1e-5
    dgm, dbt = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dc = ops.bn_bwd(g.to(dev), cdv, saved, rows, C, rows, dgm, dbt, mask_mode=2)
    assert rel_err(dc, cd.grad) < 2e-5
    assert rel_err(dgm, gd.grad) < 2e-5 and rel_err(dbt, bd.grad) < 2e-5
    # eval mode uses the running statistics
    saved_e = ops.bn_finalize(None, C, 0, gamma.to(dev), beta.to(dev), rmd, rvd, None, 0.1, 1e-5, False)
    z_e = F.batch_norm(c.double(), rm_ref, rv_ref, gamma.double(), beta.double(), training=False, eps=1e-5)
    assert rel_err(ops.bn_add_relu(cdv, saved_e, rows, C), torch.relu(z_e)) < 1e-5


def test_bottleneck_tail(dev):
    from oaprogressionmmf_amd import ops
    rows, C = 900, 256
    c3, idt, cdn = rnd(rows, C), rnd(rows, C), rnd(rows, C)
    s3 = torch.stack([rnd(C), rnd(C), rnd(C) * 0.5 + 1, rnd(C) * 0.1])
    sd = torch.stack([rnd(C), rnd(C), rnd(C) * 0.5 + 1, rnd(C) * 0.1])
    y = ops.bn_add_relu(c3.to(dev), s3.to(dev), rows, C, idt=idt.to(dev))
    assert rel_err(y, torch.relu(c3 * s3[2] + s3[3] + idt)) < 1e-6
    y = ops.bn_add_relu(c3.to(dev), s3.to(dev), rows, C, idt=cdn.to(dev), idsaved=sd.to(dev))
    assert rel_err(y, torch.relu(c3 * s3[2] + s3[3] + cdn * sd[2] + sd[3])) < 1e-6


@pytest.mark.parametrize("N,H,W", [(3, 40, 40), (2, 35, 31)])
def test_maxpool_gap(dev, N, H, W):
    from oaprogressionmmf_amd import ops
    C = 64
    c = rnd(N, C, H, W)
    sc, sh = rnd(C) * 0.5 + 1.0, rnd(C) * 0.3
    a = torch.relu(c.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).requires_grad_(True)
    y_ref = F.max_pool2d(a, 3, 2, 1)
    dy = rnd(*y_ref.shape)
    y_ref.backward(dy.double())
    saved = torch.stack([sc, sc, sc, sh]).to(dev)
    y, am = ops.maxpool_fwd(nhwc(c).to(dev), saved, N, H, W, C)
    assert rel_err(nchw(y.cpu()), y_ref) < 1e-6
    da = ops.maxpool_bwd(nhwc(dy).to(dev), am, N, H, W, C)
    assert rel_err(nchw(da.cpu()), a.grad) < 1e-6
    yy = rnd(N, 25, 2048)
    assert rel_err(ops.gap_fwd(yy.to(dev), N, 25, 2048), yy.mean(1)) < 1e-6
    dd = rnd(N, 2048)
    assert rel_err(ops.gap_bwd(dd.to(dev), N, 25, 2048), (dd / 25)[:, None, :].expand(N, 25, 2048)) < 1e-6


def test_fold_downscale(dev):
    from oaprogressionmmf_amd import ops
    B, R, Cc, S = 2, 36, 28, 25
    x = rnd(B, 1, R, Cc, S)
    out = ops.slice_fold(x.to(dev), B, R, Cc, S)
    ref = x[:, 0].permute(0, 3, 1, 2).reshape(B * S, R, Cc)
    assert torch.equal(out.cpu(), ref)
    for fs, sf in ((1, (0.5, 0.5, 1.0)), (2, (0.5, 0.5, 0.5))):
        S2 = 26
        x = rnd(B, 1, R, Cc, S2)
        ref = F.interpolate(x, scale_factor=sf, recompute_scale_factor=True, align_corners=False, mode="trilinear")
        out = ops.downscale2(x.to(dev), B, R, Cc, S2, fs)
        assert rel_err(out.reshape(ref.shape), ref) < 1e-6
    x = rnd(B, 1, 70, 50)
    ref = F.interpolate(x, scale_factor=(0.5, 0.5), recompute_scale_factor=True, align_corners=False, mode="bilinear")
    out = ops.downscale2(x.to(dev), B, 70, 50, 1, 1)
    assert rel_err(out.reshape(ref.shape), ref) < 1e-6


@pytest.mark.parametrize("rows,D", [(736, 2048), (37, 2048), (50, 64)])
def test_layernorm(dev, rows, D):
    from oaprogressionmmf_amd import ops
    x = (rnd(rows, D) * 3 + 1).double().requires_grad_(True)
    g, b = (rnd(D) * 0.5 + 1).double().requires_grad_(True), rnd(D).double().requires_grad_(True)
    y_ref = F.layer_norm(x, (D,), g, b, 1e-5)
    dy = rnd(rows, D)
    y_ref.backward(dy.double())
    xd, gd, bd = x.detach().float().to(dev), g.detach().float().to(dev), b.detach().float().to(dev)
    y, mean, rstd = ops.layernorm_fwd(xd, gd, bd, rows, D, 1e-5)
    assert rel_err(y, y_ref) < 2e-6
    dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
    dx = ops.layernorm_bwd(dy.to(dev), xd, gd, mean, rstd, dg, db, rows, D)
    assert rel_err(dx, x.grad) < 1e-5
    assert rel_err(dg, g.grad) < 1e-5 and rel_err(db, b.grad) < 1e-5


@pytest.mark.parametrize("B,n,h,d", [(2, 92, 8, 256), (2, 25, 8, 256), (3, 64, 8, 256), (1, 161, 4, 16),
                                     (2, 483, 8, 256), (1, 512, 2, 64), (1, 31, 3, 36),          # one launch (n <= 512)
                                     (1, 520, 2, 32)])                                           # past it: three launches
def test_attention(dev, B, n, h, d):
    from oaprogressionmmf_amd import ops
    dim = h * d
    scale = dim ** -0.5
    qkv = rnd(B, n, 3 * dim).double().requires_grad_(True)
    q, k, v = qkv.reshape(B, n, 3, h, d).permute(2, 0, 3, 1, 4)
    dots = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    attn_ref = dots.softmax(-1)
    out_ref = torch.einsum("bhij,bhjd->bhid", attn_ref, v).permute(0, 2, 1, 3).reshape(B, n, dim)
    dout = rnd(B, n, dim)
    out_ref.backward(dout.double())
    qd = qkv.detach().float().to(dev)
    out, attn = ops.attention_fwd(qd, B, n, h, d, scale)
    assert rel_err(attn, attn_ref) < 2e-6
    assert rel_err(out, out_ref) < 2e-6
    dqkv = ops.attention_bwd(dout.to(dev), qd, attn, B, n, h, d, scale)
    assert rel_err(dqkv, qkv.grad) < 2 * BWD


def test_pointwise_loss_adam(dev):
    from oaprogressionmmf_amd import ops
    x = rnd(1000, 37) * 2
    xd = x.double().requires_grad_(True)
    y_ref = F.gelu(xd)
    dy = rnd(1000, 37)
    y_ref.backward(dy.double())
    assert rel_err(ops.gelu_fwd(x.to(dev)), y_ref) < 1e-6
    assert rel_err(ops.gelu_bwd(dy.to(dev), x.to(dev)), xd.grad) < 1e-5
    assert rel_err(ops.relu_fwd(x.to(dev)), torch.relu(x)) == 0
    assert rel_err(ops.add(x.to(dev), dy.to(dev)), x + dy) == 0
    # dropout: keep fraction and scaling
    big = torch.ones(1 << 20, device=dev)
    d1 = ops.dropout(big, 0.1, 12345)
    keep = (d1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 2e-3
    assert abs(d1.max().item() - 1 / 0.9) < 1e-6
    assert torch.equal(d1, ops.dropout(big, 0.1, 12345)) and not torch.equal(d1, ops.dropout(big, 0.1, 999))
    # focal loss incl. extreme logits
    logits = torch.cat([rnd(60, 2) * 3, torch.tensor([[30.0, -30.0], [-30.0, 30.0], [0.0, 0.0], [50.0, 50.0]])])
    tgt = torch.randint(0, 2, (64,), generator=G)
    ld = logits.double().requires_grad_(True)
    logpt = -F.cross_entropy(ld, tgt, reduction="none")
    pt = torch.exp(logpt)
    loss_ref = (-((1 - pt) ** 2.0) * logpt).mean()
    loss_ref.backward()
    loss, dl = ops.focal_loss(logits.to(dev), tgt.to(dev), 2.0)
    assert abs(loss.item() - loss_ref.item()) < 1e-6 * max(1, abs(loss_ref.item()))
    assert rel_err(dl, ld.grad) < 1e-5
    loss, dl = ops.focal_loss(logits.to(dev), tgt.to(dev), 0.0, focal=False)
    ld.grad = None
    ce = F.cross_entropy(ld, tgt)
    ce.backward()
    assert abs(loss.item() - ce.item()) < 1e-6 * max(1, ce.item()) and rel_err(dl, ld.grad) < 1e-5
    # Adam, 3 steps against torch.optim.Adam
    n = 10007
    p0, g = rnd(n), [rnd(n) for _ in range(3)]
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=1e-3, weight_decay=1e-4)
    pd = torch.zeros(n + 5, device=dev)[:n]
    pd.copy_(p0)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for s in range(3):
        pr.grad = g[s].clone()
        opt.step()
        ops.adam_step(pd, g[s].to(dev), m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, s + 1)
    assert rel_err(pd, pr.detach()) < 1e-6


@pytest.mark.parametrize("case", [(3, 20, 20, 64, 64, 3, 1, 1), (2, 21, 19, 128, 128, 3, 2, 1), (2, 20, 20, 256, 512, 1, 2, 0),
                                  (3, 20, 20, 256, 64, 1, 1, 0)])
@pytest.mark.parametrize("mode,second", [(1, True), (1, False), (2, False)])
def test_conv2d_dgrad_fused_bn_backward(dev, case, mode, second):
    """dgrad epilogue = residual add + ReLU mask + BatchNorm-backward column sums (one or two BatchNorms)"""
    from oaprogressionmmf_amd import ops
    N, H, W, Cin, Cout, k, s, p = case
    OH, OW = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
    w = rnd(Cout, Cin, k, k, scale=(Cin * k * k) ** -0.5)
    dy, res = rnd(N, OH, OW, Cout), rnd(N, H, W, Cin)
    c, y, c2 = rnd(N, H, W, Cin), rnd(N, H, W, Cin), rnd(N, H, W, Cin)
    sv = torch.stack([rnd(Cin) * 0.3, torch.rand(Cin, generator=G) + 0.5, rnd(Cin) * 0.5 + 1, rnd(Cin) * 0.2])
    sv2 = torch.stack([rnd(Cin) * 0.3, torch.rand(Cin, generator=G) + 0.5, rnd(Cin), rnd(Cin)])
    wp = packw(w).to(dev)
    g = ops.conv2d_dgrad(dy.to(dev), wp, N, H, W, Cin, Cout, k, k, s, p, residual=res.to(dev)).cpu().double()
    mask = (y > 0) if mode == 1 else ((c * sv[2] + sv[3]) > 0)
    dz_ref = g * mask
    xh = (c.double() - sv[0].double()) * sv[1].double()
    sums = [dz_ref.sum((0, 1, 2)), (dz_ref * xh).sum((0, 1, 2))]
    bnb = dict(mode=mode, c=c.to(dev), saved=sv.to(dev), y=y.to(dev) if mode == 1 else None)
    if second:
        bnb["c2"], bnb["saved2"] = c2.to(dev), sv2.to(dev)
        sums.append((dz_ref * ((c2.double() - sv2[0].double()) * sv2[1].double())).sum((0, 1, 2)))
    dz, part = ops.conv2d_dgrad(dy.to(dev), wp, N, H, W, Cin, Cout, k, k, s, p, residual=res.to(dev), bnb=bnb)
    assert rel_err(dz, dz_ref) < BWD
    assert part.shape[1] == len(sums)
    for i, ref in enumerate(sums):
        assert rel_err(part[:, i].double().sum(0), ref) < 2e-5, i


def test_batch_augment_vs_reference(dev):
    """device input pipeline (PTBatchAugment: unit range -> in-plane rotation -> gamma -> normalise, one fused kernel)
    against the reference's own transform classes, fixture F12 (pinned random states); 2e-5 absolute on values of
    order 1 (bilinear weights and powf in fp32)"""
    from oaprogressionmmf_amd.preproc import PTBatchAugment
    g = load("f12_augment.npz")
    states = [tuple(s) for s in g["states"]]
    for tag, shape in (("mr", (4, 1, 24, 20, 6)), ("xr", (4, 1, 28, 22))):
        raw = (np.abs(P.make_input("aug_" + tag, shape)) * 300.0 + 5.0).astype(np.float32)
        mean, std = g[tag + ":norm"]
        aug = PTBatchAugment(mean=[float(mean)], std=[float(std)])
        y = aug(torch.from_numpy(raw).to(dev), states=states).cpu().numpy()
        assert y.shape == g[tag].shape
        assert np.abs(y - g[tag]).max() < 2e-5, tag
    # validation / test pipeline (no random parts) == unit range + normalise, and the drawn-state path runs
    ev = PTBatchAugment(mean=0.5, std=0.25, rotate_prob=0.0, gamma_prob=0.0)
    x = torch.from_numpy(raw).to(dev)
    want = ((raw - raw.reshape(4, -1).min(1)[:, None, None, None]) /
            (raw.reshape(4, -1).max(1) - raw.reshape(4, -1).min(1))[:, None, None, None] - 0.5) / 0.25
    assert np.abs(ev(x).cpu().numpy() - want).max() < 1e-6
    out = PTBatchAugment(mean=0.5, std=0.25)(x)
    assert out.shape == x.shape and torch.isfinite(out).all()
    with pytest.raises(ValueError):
        aug(x[:, 0])


def test_integer_volume_upload_and_prefetch(dev):
    """(f-3) the loader ships the volumes as stored on disk -- uint8 radiographs, uint16 / int16 MRI -- and the device
    widens them (koaf_widen): the same pipeline output as from the fp32 tensor, bit for bit; PinnedPrefetcher stages the
    next batch in pinned memory on its own stream and hands over identical device tensors in loader order"""
    from oaprogressionmmf_amd import ops
    from oaprogressionmmf_amd.preproc import PTBatchAugment, PinnedPrefetcher
    rng = np.random.default_rng(3)
    for dt, hi, shape in ((torch.uint8, 255, (3, 1, 30, 22)), (torch.uint16, 4095, (2, 1, 24, 20, 7)), (torch.int16, 3000, (2, 1, 17, 13, 5))):
        raw = torch.from_numpy(rng.integers(0, hi, size=shape).astype(np.int64)).to(dt)
        assert torch.equal(ops.widen(raw.to(dev)).cpu(), raw.float())
        aug = PTBatchAugment(mean=0.4, std=0.2)
        states = aug.draw(shape[0])
        assert torch.equal(aug(raw.to(dev), states=states), aug(raw.float().to(dev), states=states))
    batches = [{"image__xr_pa": torch.from_numpy(rng.integers(0, 255, size=(2, 1, 16, 16)).astype(np.uint8)),
                "image__sag_3d_dess": torch.from_numpy(rng.integers(0, 4000, size=(2, 1, 8, 8, 4)).astype(np.uint16)),
                "target": torch.tensor([[i], [i + 1]]), ("-", "exam_knee_id"): [f"k{i}", f"k{i}b"]} for i in range(5)]
    seen = 0
    for got, want in zip(PinnedPrefetcher(batches, dev), batches):
        for k, v in want.items():
            if torch.is_tensor(v):
                assert got[k].is_cuda and got[k].dtype == v.dtype and torch.equal(got[k].cpu(), v), k
            else:
                assert got[k] == v
        seen += 1
    assert seen == len(batches)


def test_dropout2d_channels(dev):
    """koaf_dropout2d: whole (image, channel) planes are dropped or scaled by 1/(1-p); the same draws as the
    element-wise generator makes on the pooled (N, C) tensor with the same seed; backward applies the same mask"""
    from oaprogressionmmf_amd import functional as KF, ops
    N, C, h, w, p = 6, 64, 3, 5, 0.3
    x = (torch.rand(N, h, w, C, generator=G) + 0.5).to(dev)
    seed = 12345
    y = ops.dropout2d(x, N, h * w, C, p, seed)
    ratio = (y / x).cpu()
    per_nc = ratio.permute(0, 3, 1, 2).reshape(N, C, -1)
    assert torch.all((per_nc - per_nc[:, :, :1]).abs() < 1e-6), "a channel plane is not uniformly kept / dropped"
    keep = per_nc[:, :, 0]
    assert torch.all((keep == 0) | ((keep - 1 / (1 - p)).abs() < 1e-5))
    assert 0.15 < (keep == 0).float().mean().item() < 0.45
    pooled = ops.dropout(torch.ones(N, C, device=dev), p, seed).cpu()
    assert torch.equal((pooled == 0), (keep == 0))
    # autograd path on the (N, C, h, w) view
    xv = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    torch.manual_seed(0)
    out = KF.dropout2d(xv, p, True)
    out.sum().backward()
    assert torch.equal((xv.grad == 0), (out == 0)) and out.shape == xv.shape
    assert KF.dropout2d(xv, p, False) is xv


def test_shifted_batchnorm_statistics(dev):
    """BatchNorm batch statistics of a conv output whose channel means are hundreds of standard deviations away from
    zero: summed about a shift near the mean (the running mean, KoafGemm.stats_shift) the variance keeps fp32-level
    accuracy; unshifted, E[x^2] - mean^2 loses most of its digits.  Ragged last tile and a 3x3 gather included."""
    from oaprogressionmmf_amd import ops
    N, H, W, Cin, Cout = 3, 19, 17, 64, 64
    x = torch.rand(N, Cin, H, W, generator=G) * 0.02 + 1.0                 # nearly constant, positive input
    w = rnd(Cout, Cin, 3, 3, scale=0.01) + 0.05                            # weights with a common positive part
    y_ref = F.conv2d(x.double(), w.double(), padding=1)
    mean_ref = y_ref.mean((0, 2, 3))
    var_ref = y_ref.var((0, 2, 3), unbiased=False)
    assert (mean_ref.abs() / var_ref.sqrt()).median() > 3                  # (border pixels keep the std from vanishing)
    rows = N * H * W
    xd, wp = nhwc(x).to(dev), packw(w).to(dev)
    gamma, beta = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    errs = {}
    for tag, shift in (("none", None), ("near", (mean_ref * 0.98).float().to(dev))):
        y, part = ops.conv2d_fwd(xd, wp, N, H, W, Cin, Cout, 3, 3, 1, 1, stats=True, shift=shift)
        rm = shift.clone() if shift is not None else torch.zeros(Cout, device=dev)
        rv, nbt = torch.ones(Cout, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
        saved = ops.bn_finalize(part, Cout, rows, gamma, beta, rm, rv, nbt, 0.1, 1e-5, True, shift=rm if shift is not None else None)
        mean, invstd = saved[0].double().cpu(), saved[1].double().cpu()
        errs[tag] = (rel_err(mean, mean_ref), rel_err(invstd, 1.0 / torch.sqrt(var_ref + 1e-5)))
        # the running mean is updated from its old value (which doubled as the shift): 0.9 * old + 0.1 * batch mean
        want_rm = 0.9 * (shift.double().cpu() if shift is not None else torch.zeros(Cout).double()) + 0.1 * mean_ref
        assert rel_err(rm, want_rm) < 1e-6
    assert errs["near"][0] < 1e-6 and errs["near"][1] < 2e-6, errs
    assert errs["near"][1] <= errs["none"][1], errs


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,p", [(3, 19, 17, 64, 128, 3, 1, 1), (2, 20, 20, 128, 64, 3, 2, 1),
                                                  (5, 13, 11, 256, 64, 1, 1, 0), (2, 14, 14, 64, 256, 1, 2, 0),
                                                  (1, 9, 9, 512, 512, 3, 1, 1), (13, 64, 64, 64, 128, 1, 1, 0)])
def test_fp16_scheme_weight_images_bit_identical(dev, N, H, W, Cin, Cout, k, s, p):
    """conv forward (+BatchNorm prologue, statistics) and data gradient (+residual) with the weight tiles DMA'd from the
    plane images (koaf_wplanes_build -> global_load_lds) = the same bits as with the fp32 weight cut inside the kernel at
    the same scale (wimg without images)"""
    from oaprogressionmmf_amd import ops
    x = rnd(N, H, W, Cin).to(dev)
    w = rnd(Cout, k, k, Cin, scale=(k * k * Cin) ** -0.5).to(dev)
    sc, sh = (rnd(Cin) * 0.2 + 1).to(dev), (rnd(Cin) * 0.1).to(dev)
    img = ops.build_weight_planes(w, Cout, k * k, Cin)
    assert float(img[2]) == float(w.abs().max())
    noimg = (None, None, img[2])
    for tf in (False, True):
        a = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, k, k, s, p, sc if tf else None, sh if tf else None, stats=True, wimg=noimg)
        b = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, k, k, s, p, sc if tf else None, sh if tf else None, stats=True, wimg=img,
                           aplanes=False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    OH, OW = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
    dy, res = rnd(N, OH, OW, Cout).to(dev), rnd(N, H, W, Cin).to(dev)
    am = amax_of(dy)
    a = ops.conv2d_dgrad(dy, w, N, H, W, Cin, Cout, k, k, s, p, residual=res, wimg=noimg, dy_amax=am)
    b = ops.conv2d_dgrad(dy, w, N, H, W, Cin, Cout, k, k, s, p, residual=res, wimg=img, dy_amax=am, aplanes=False)
    assert torch.equal(a, b)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,p", [(3, 19, 17, 64, 64, 3, 1, 1), (2, 20, 20, 128, 128, 3, 2, 1), (7, 24, 24, 64, 128, 3, 1, 1),
                                                  (2, 9, 31, 256, 64, 3, 1, 1), (1, 5, 5, 32, 64, 3, 1, 1), (2, 21, 21, 64, 64, 3, 2, 1),
                                                  (1, 7, 120, 64, 64, 3, 1, 1), (3, 96, 96, 64, 64, 3, 1, 1), (5, 12, 12, 512, 256, 3, 1, 1),
                                                  (2, 48, 48, 128, 128, 3, 1, 1), (1, 8, 16, 64, 64, 3, 1, 1), (3, 24, 32, 128, 64, 3, 1, 1),
                                                  (2, 16, 16, 64, 128, 3, 1, 1)])
def test_activation_plane_images(dev, N, H, W, Cin, Cout, k, s, p):
    """3x3 conv forward with the INPUT tiles DMA'd from activation plane images (koaf_act_planes: BatchNorm + ReLU prologue
    cut once into fp16 piece planes).  Per-tap gather kernel (global_load_lds of one pixel's 8 channels per lane, padding
    from the zero chunk): the same bits -- output and BatchNorm statistics -- as the fp32 loader that converts every element
    once per filter tap.  Halo kernel (stride 1, rows <= 96 pixels: 256-pixel raster tiles, all nine taps read from one
    LDS-resident pixel range, off-image taps masked in registers): the same sums reassociated, so equal to fp32 rounding
    and equally close to float64.  Ragged last tiles, stride 2, images smaller than a tile, rows too wide for the halo.
    64- / 128-channel layers whose images are whole 8 x 16 pixel tiles run the rectangle-tile kernel (M_PT: zero-filled halo, weight
    fragments straight into registers) under the per-layer choice (mode 1): the last four shapes and 96 x 96."""
    from oaprogressionmmf_amd import ops
    x = rnd(N, H, W, Cin).to(dev)
    w = rnd(Cout, k, k, Cin, scale=(k * k * Cin) ** -0.5).to(dev)
    sc, sh = (rnd(Cin) * 0.2 + 1).to(dev), (rnd(Cin) * 0.1).to(dev)
    img = ops.build_weight_planes(w, Cout, k * k, Cin)
    for tf in (False, True):
        args = (x, w, N, H, W, Cin, Cout, k, k, s, p, sc if tf else None, sh if tf else None)
        a = ops.conv2d_fwd(*args, stats=True, wimg=img, aplanes=False)
        was = ops.set_conv3x3_halo(False)
        try:
            b = ops.conv2d_fwd(*args, stats=True, wimg=img, aplanes=True)
        finally:
            ops.set_conv3x3_halo(was)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        xin = x.double().cpu()
        if tf:
            xin = torch.relu(xin * sc.double().cpu() + sh.double().cpu())
        ref = F.conv2d(xin.permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), stride=s, padding=p).permute(0, 2, 3, 1)
        ea = rel_err(a[0].cpu(), ref)
        for mode in (1, 2, 3):                    # per-layer choice, 256-row shape, 128-row shape (two blocks per CU)
            was = ops.set_conv3x3_halo(mode)
            try:
                c = ops.conv2d_fwd(*args, stats=True, wimg=img, aplanes=True)
            finally:
                ops.set_conv3x3_halo(was)
            ec = rel_err(c[0].cpu(), ref)
            assert ec < 2e-6 and ec < 1.5 * ea + 1e-7, (mode, ea, ec)
            assert rel_err(c[0].cpu(), a[0].cpu().double()) < 1e-6
            # statistics: per-tile partial rows differ in count (256- / 128-row tiles), their column totals agree
            assert rel_err(c[1].sum(0).cpu(), a[1].sum(0).cpu().double()) < 1e-5
    # the images themselves: hi + lo == clamp(relu(sc * x + sh) * 16) to 2^-24 relative, zero chunk behind them
    pl = ops.act_planes(x, N * H * W, Cin, 1, sc, sh, fscale=ops.ACT_SCALE)
    n = N * H * W * Cin
    hi, lo = pl[:n].view(torch.float16).double(), pl[n:2 * n].view(torch.float16).double()
    ref = (torch.relu(x.double() * sc.double() + sh.double()) * 16.0).clamp(max=65504.0).flatten()
    assert float(((hi + lo) - ref).abs().max() / ref.abs().max()) < 2.0 ** -22
    assert int(pl[2 * n:].abs().max()) == 0


@pytest.mark.parametrize("gs,ws", [(1e-9, 1e-4), (3e4, 50.0), (1.0, 1.0)])
def test_fp16_scheme_scales(dev, gs, ws):
    """operands far from unit scale: tiny gradients (1e-9), weights of magnitude 1e-4 / 50, and a gradient tensor whose
    rows span eight decades -- the scale taken from the tensor's amax keeps every contraction at fp32 rounding level"""
    from oaprogressionmmf_amd import ops
    N, H, W, Cin, Cout, k = 4, 12, 12, 128, 128, 3
    x = rnd(N, Cin, H, W)
    w = rnd(Cout, Cin, k, k, scale=ws * (Cin * k * k) ** -0.5)
    dy = rnd(N, Cout, H, W) * gs
    dy *= 10.0 ** (-8.0 * torch.rand(N, 1, H, W, generator=G))           # per-pixel magnitudes over eight decades
    xin, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y_ref = F.conv2d(xin, wd, padding=1)
    y_ref.backward(dy.double())
    xd, wp, dyd = nhwc(x).to(dev), packw(w).to(dev), nhwc(dy).to(dev)
    img, am = ops.build_weight_planes(wp, Cout, k * k, Cin), amax_of(nhwc(dy).to(dev))
    y, _ = ops.conv2d_fwd(xd, wp, N, H, W, Cin, Cout, k, k, 1, 1, wimg=img)
    assert rel_err(nchw(y.cpu()), y_ref) < 2e-6
    dx = ops.conv2d_dgrad(dyd, wp, N, H, W, Cin, Cout, k, k, 1, 1, wimg=img, dy_amax=am)
    assert rel_err(nchw(dx.cpu()), xin.grad) < BWD
    # pixels whose gradient is eight decades under the tensor's amax: each input-gradient pixel still has its own digits
    err_px = ((nchw(dx.cpu()).double() - xin.grad).flatten(1).norm(dim=1) / xin.grad.flatten(1).norm(dim=1))
    assert float(err_px.max()) < 1e-5
    dw = torch.empty(Cout, k, k, Cin, device=dev)
    ops.conv2d_wgrad(dyd, xd, dw, N, H, W, Cin, Cout, k, k, 1, 1, dy_amax=am)
    assert rel_err(dw.cpu().permute(0, 3, 1, 2), wd.grad) < BWD


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,p", [(3, 19, 17, 64, 128, 3, 1, 1), (2, 20, 20, 128, 64, 3, 2, 1),
                                                  (5, 13, 11, 256, 64, 1, 1, 0), (2, 14, 14, 64, 256, 1, 2, 0)])
def test_bn_backward_apply_formed_in_the_gemm_loaders(dev, N, H, W, Cin, Cout, k, s, p):
    """ops.BnApply: dc = coef0*dz + coef3 - coef2*c evaluated on load by the dgrad / wgrad GEMMs (KoafOperand.tf 2) == the
    same convolutions on the materialised dc (koaf_bn_bwd_apply), to fp32 rounding of the different evaluation order, and
    both == the float64 BatchNorm + convolution backward; the scale bound koaf_bn_bwd_finalize leaves is >= max |dc|"""
    from oaprogressionmmf_amd import ops
    OH, OW = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
    rows = N * OH * OW
    x = rnd(N, Cin, H, W)
    w = rnd(Cout, Cin, k, k, scale=(Cin * k * k) ** -0.5)
    c = rnd(N, Cout, OH, OW) * 1.5 + 0.3                   # the conv output whose BatchNorm is back-propagated
    g = rnd(N, Cout, OH, OW) * 1e-3                        # gradient w.r.t. relu(bn(c))
    gamma, beta = rnd(Cout) * 0.2 + 1.0, rnd(Cout) * 0.1
    # float64 reference: y = relu(bn_train(c)); dc = d/dc; then dx, dw of conv(x, w) given dc
    cd = c.double().requires_grad_(True)
    yb = torch.relu(F.batch_norm(cd, None, None, gamma.double(), beta.double(), True, 0.0, 1e-5))
    (yb * g.double()).sum().backward()
    dc_ref = cd.grad
    xd64, wd64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xd64, wd64, stride=s, padding=p).backward(dc_ref)
    # device: statistics of c, then the fused recipe
    cdv, gdv = nhwc(c).to(dev), nhwc(g).to(dev)
    rm, rv, nbt = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    saved = ops.bn_finalize(ops.colstats(cdv, rows, Cout), Cout, rows, gamma.to(dev), beta.to(dev), rm, rv, nbt, 0.1, 1e-5, True)
    dg, db = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    ap = ops.bn_bwd(gdv.clone(), cdv, saved, rows, Cout, rows, dg, db, 2, fused=True)
    assert isinstance(ap, ops.BnApply)
    dc = ap.materialize(want_amax=True)
    assert rel_err(nchw(dc.cpu()), dc_ref) < 1e-5
    assert float(ap.amax) >= float(dc._koaf_amax) > 0        # a bound of the largest magnitude, never under it
    assert float(ap.amax) < 3e4 * float(dc._koaf_amax)       # ... and within the range the fp16 pieces have to spare
    xdv, wp = nhwc(x).to(dev), packw(w).to(dev)
    img = ops.build_weight_planes(wp, Cout, k * k, Cin)
    dx_f = ops.conv2d_dgrad(ap, wp, N, H, W, Cin, Cout, k, k, s, p, wimg=img, aplanes=False)
    dx_m = ops.conv2d_dgrad(dc, wp, N, H, W, Cin, Cout, k, k, s, p, wimg=img, dy_amax=dc._koaf_amax, aplanes=False)
    assert rel_err(dx_f, dx_m.double()) < 2e-6
    assert rel_err(nchw(dx_f.cpu()), xd64.grad) < 1e-5
    if k > 1:
        # the same gradients with dy cut once into activation plane images (apply included) and gathered by LDS-DMA: the
        # per-tap gather kernel (also the four parity classes of stride 2) reproduces the loader's bits, the halo kernel
        # (stride 1) reassociates the sums; with a residual and the fused BatchNorm-backward epilogue of the producer
        res = rnd(N, H, W, Cin).to(dev)
        was = ops.set_conv3x3_halo(False)
        try:
            dx_g = ops.conv2d_dgrad(ap, wp, N, H, W, Cin, Cout, k, k, s, p, wimg=img, aplanes=True)
            dx_gm = ops.conv2d_dgrad(dc, wp, N, H, W, Cin, Cout, k, k, s, p, wimg=img, dy_amax=dc._koaf_amax, aplanes=True)
        finally:
            ops.set_conv3x3_halo(was)
        assert torch.equal(dx_g, dx_f) and torch.equal(dx_gm, dx_m)
        dx_h = ops.conv2d_dgrad(ap, wp, N, H, W, Cin, Cout, k, k, s, p, wimg=img, aplanes=True)
        assert rel_err(dx_h, dx_f.double()) < 1e-6 and rel_err(nchw(dx_h.cpu()), xd64.grad) < 1e-5
        cx = rnd(N, H, W, Cin).to(dev) * 1.5 + 0.3           # the producer's conv output and its BatchNorm
        savx = ops.bn_finalize(ops.colstats(cx, N * H * W, Cin), Cin, N * H * W, (rnd(Cin) * 0.2 + 1).to(dev), (rnd(Cin) * 0.1).to(dev),
                               torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
                               0.1, 1e-5, True)
        bnb = dict(mode=2, c=cx, saved=savx, dz_amax=True)
        r0 = ops.conv2d_dgrad(ap, wp, N, H, W, Cin, Cout, k, k, s, p, residual=res, bnb=bnb, wimg=img, aplanes=False)
        r1 = ops.conv2d_dgrad(ap, wp, N, H, W, Cin, Cout, k, k, s, p, residual=res, bnb=bnb, wimg=img, aplanes=True)
        assert rel_err(r1[0], r0[0].double()) < 1e-6 and rel_err(r1[1].sum(0), r0[1].sum(0).double()) < 1e-4
        assert abs(float(r1[2]) / float(r0[2]) - 1) < 1e-5
    dw_f, dw_m = torch.empty(Cout, k, k, Cin, device=dev), torch.empty(Cout, k, k, Cin, device=dev)
    ops.conv2d_wgrad(ap, xdv, dw_f, N, H, W, Cin, Cout, k, k, s, p, aplanes=False)
    ops.conv2d_wgrad(dc, xdv, dw_m, N, H, W, Cin, Cout, k, k, s, p, dy_amax=dc._koaf_amax, aplanes=False)
    assert rel_err(dw_f, dw_m.double()) < 2e-6
    assert rel_err(dw_f.cpu().permute(0, 3, 1, 2), wd64.grad) < 1e-5
    # weight gradient with BOTH operands from activation plane images, K-major by LDS-DMA (with and without the BatchNorm
    # prologue of x; 1x1 kernels as the one-tap gather; stride 2): the loaders' bits, so the same sums in the same order --
    # except 3x3 / stride 1, which runs the padded-raster ring kernel (koaf_wgrad3.hip): same products, sums reassociated
    scx, shx = (rnd(Cin) * 0.2 + 1).to(dev), (rnd(Cin) * 0.1).to(dev)
    ring = k == 3 and s == 1

    def same(a, b):
        return rel_err(a, b.double()) < 1e-6 if ring else torch.equal(a, b)
    for tf in (False, True):
        a0, a1 = torch.empty(Cout, k, k, Cin, device=dev), torch.empty(Cout, k, k, Cin, device=dev)
        ops.conv2d_wgrad(ap, xdv, a0, N, H, W, Cin, Cout, k, k, s, p, scx if tf else None, shx if tf else None, aplanes=False)
        ops.conv2d_wgrad(ap, xdv, a1, N, H, W, Cin, Cout, k, k, s, p, scx if tf else None, shx if tf else None, aplanes=True)
        assert same(a1, a0)
    ops.conv2d_wgrad(dc, xdv, a1, N, H, W, Cin, Cout, k, k, s, p, scx, shx, dy_amax=dc._koaf_amax, aplanes=True)
    ops.conv2d_wgrad(dc, xdv, a0, N, H, W, Cin, Cout, k, k, s, p, scx, shx, dy_amax=dc._koaf_amax, aplanes=False)
    assert same(a1, a0)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(3, 16, 16, 64, 64), (2, 24, 24, 128, 64), (5, 17, 19, 64, 128), (1, 48, 48, 64, 64),
                                            (2, 31, 96, 64, 64), (300, 16, 16, 64, 64), (2, 20, 16, 256, 192),
                                            (3, 12, 12, 64, 64)])          # (W < 16: the generic K-major kernel)
def test_wgrad3x3_ring_kernel(dev, N, H, W, Cin, Cout):
    """3x3 / stride 1 / pad 1 weight gradient over plane images in padded raster order (koaf_wgrad3.hip: x in an LDS ring, all
    nine taps from one walk): against float64 autograd at the gradient bar; image borders (every tap that leaves the image is
    a zero), rows narrower than a 32-position chunk, several k-ranges and (co, ci) tiles, a BatchNorm prologue on x"""
    from oaprogressionmmf_amd import ops
    x = rnd(N, Cin, H, W)
    sc, sh = rnd(Cin) * 0.2 + 1, rnd(Cin) * 0.1
    dy = rnd(N, Cout, H, W) * 1e-2
    a = torch.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    wd = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(a, wd, padding=1).backward(dy.double())
    xd, dyd = nhwc(x).to(dev), nhwc(dy).to(dev)
    dw = torch.full((Cout, 3, 3, Cin), float("nan"), device=dev)
    ops.conv2d_wgrad(dyd, xd, dw, N, H, W, Cin, Cout, 3, 3, 1, 1, sc.to(dev), sh.to(dev), dy_amax=amax_of(dyd), aplanes=True)
    assert rel_err(dw.cpu().permute(0, 3, 1, 2), wd.grad) < BWD
    # every tap on its own (a border bug would hide in the norm of the whole tensor)
    for t in range(9):
        assert rel_err(dw.cpu()[:, t // 3, t % 3, :], wd.grad[:, :, t // 3, t % 3]) < BWD, t





def test_numerics_status_words_saturation_and_nonfinite(dev):
    """The fp16 contraction scheme clamps its scaled pieces to +-65504: behind the FIXED activation scale 16 an activation
    beyond 4094 saturates, and a NaN piece becomes a clamp bound.  Neither may stay silent (koaf.h koaf_set_status_buffer):
    (1) saturated activations are counted by every producer / loader of a fixed-scale operand;
    (2) a NaN in a weight or gradient reaches that operand's amax scalar and the GEMM returns NaN everywhere;
    (3) non-finite BatchNorm coefficients are counted.  Healthy inputs flag nothing."""
    from oaprogressionmmf_amd import ops
    N, H, W, Cin, Cout = 2, 12, 12, 64, 128
    rows = N * H * W
    x = rnd(N, H, W, Cin).to(dev)
    w1 = rnd(Cout, 1, 1, Cin, scale=0.1).to(dev)
    w3 = rnd(Cout, 3, 3, Cin, scale=0.05).to(dev)
    img1, img3 = ops.build_weight_planes(w1, Cout, 1, Cin), ops.build_weight_planes(w3, Cout, 9, Cin)
    sc, sh = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    ops.numerics_status(reset=True)
    y_ok, _ = ops.conv2d_fwd(x, w1, N, H, W, Cin, Cout, 1, 1, 1, 0, sc, sh, wimg=img1)
    ops.conv2d_fwd(x, w3, N, H, W, Cin, Cout, 3, 3, 1, 1, sc, sh, wimg=img3)
    saved = torch.stack([torch.zeros(Cin), torch.ones(Cin), torch.ones(Cin), torch.zeros(Cin)]).to(dev)
    ops.bn_add_relu(x, saved, rows, Cin, idt=x)
    assert ops.numerics_status() == {"saturated": 0, "nonfinite": 0}
    # (1) relu(5000 * x) passes 4094 for about 40 % of the elements
    big = torch.full((Cin,), 5000.0, device=dev)
    y_sat, _ = ops.conv2d_fwd(x, w1, N, H, W, Cin, Cout, 1, 1, 1, 0, big, sh, wimg=img1)       # fp32 loader: per tile
    s1 = ops.numerics_status(reset=True)
    assert s1["saturated"] > 0 and s1["nonfinite"] == 0 and torch.isfinite(y_sat).all()
    ops.conv2d_fwd(x, w3, N, H, W, Cin, Cout, 3, 3, 1, 1, big, sh, wimg=img3)                  # plane images: per element
    s2 = ops.numerics_status(reset=True)
    n_over = int((torch.relu(x * 5000.0) > 4094.0).sum())
    assert s2["saturated"] == n_over, (s2, n_over)
    saved_big = saved.clone()
    saved_big[2] = 5000.0
    yb = ops.bn_add_relu(x, saved_big, rows, Cin)                                               # the bottleneck tail's output
    s3 = ops.numerics_status(reset=True)
    assert s3["saturated"] == int((yb > 4094.0).sum()) > 0
    with pytest.warns(RuntimeWarning, match="clamped"):
        ops.bn_add_relu(x, saved_big, rows, Cin)
        ops.check_numerics()
    assert ops.numerics_status() == {"saturated": 0, "nonfinite": 0}                           # check_numerics() resets
    # (2) one NaN in a weight: its plane images carry a NaN amax, every output element of the convolution is NaN
    wn = w1.clone()
    wn[3, 0, 0, 5] = float("nan")
    imgn = ops.build_weight_planes(wn, Cout, 1, Cin)
    assert torch.isnan(imgn[2]).all()
    yn, _ = ops.conv2d_fwd(x, wn, N, H, W, Cin, Cout, 1, 1, 1, 0, sc, sh, wimg=imgn)
    assert torch.isnan(yn).all() and ops.numerics_status(reset=True)["nonfinite"] >= 1
    # ... and one Inf in a gradient: max |dz| -> the bound of |dc| -> the data gradient
    g = rnd(N, H, W, Cout, scale=1e-3).to(dev)
    g[1, 2, 3, 4] = float("inf")
    c = rnd(N, H, W, Cout).to(dev)
    gam, bet = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    rm, rv, nbt = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    sv = ops.bn_finalize(ops.colstats(c, rows, Cout), Cout, rows, gam, bet, rm, rv, nbt, 0.1, 1e-5, True)
    dg, db = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    ap = ops.bn_bwd(g, c, sv, rows, Cout, rows, dg, db, 0, fused=True)
    assert not torch.isfinite(ap.amax).all()
    dx = ops.conv2d_dgrad(ap, w1, N, H, W, Cin, Cout, 1, 1, 1, 0, wimg=img1)
    assert torch.isnan(dx).all() and ops.numerics_status(reset=True)["nonfinite"] >= 1
    # (3) a NaN in a conv output poisons its channel's statistics: the BatchNorm coefficients are counted
    cn = c.clone()
    cn[0, 0, 0, 7] = float("nan")
    ops.bn_finalize(ops.colstats(cn, rows, Cout), Cout, rows, gam, bet, rm.clone(), rv.clone(), nbt, 0.1, 1e-5, True)
    assert ops.numerics_status(reset=True)["nonfinite"] == 1
    # healthy again
    y2, _ = ops.conv2d_fwd(x, w1, N, H, W, Cin, Cout, 1, 1, 1, 0, sc, sh, wimg=img1)
    assert torch.equal(y2, y_ok) and ops.numerics_status() == {"saturated": 0, "nonfinite": 0}


@pytest.mark.parametrize("ds", [False, True])
@pytest.mark.parametrize("C,Cout,bf16", [(256, 64, False), (1024, 256, False), (512, 512, False), (256, 64, True), (1024, 256, True)])
def test_bottleneck_tail_formed_in_the_conv1_loader(dev, C, Cout, bf16, ds):
    """y = relu(bn3(c3) + identity) (_torchvision.py:132-136) formed by the NEXT block's conv1 while it loads its operand
    (koaf.h KoafOperand.tf 3) and written once by the first column tile: the side-stored y and the convolution's output and
    statistics equal, bit for bit, the element-wise tail pass followed by the plain convolution -- also with several column
    tiles (Cout 256 / 512), ragged row tiles and, in the bf16 storage mode, with y rounded before it is multiplied.
    ds: the block had a downsample branch -- the identity is BatchNorm(raw downsample conv output) (_torchvision.py:129-130),
    formed on load as well"""
    from oaprogressionmmf_amd import ops
    N, H, W = 3, 21, 19
    rows = N * H * W
    c3, idt = rnd(N, H, W, C).to(dev), (rnd(N, H, W, C) if ds else torch.relu(rnd(N, H, W, C))).to(dev)
    if bf16:
        c3, idt = c3.bfloat16(), idt.bfloat16()
    saved = torch.stack([0.1 * rnd(C), 1.0 + 0.1 * rnd(C), 1.0 + 0.1 * rnd(C), 0.1 * rnd(C)]).to(dev)
    idsaved = torch.stack([0.1 * rnd(C), 1.0 + 0.1 * rnd(C), 1.0 + 0.3 * rnd(C), 0.2 * rnd(C)]).to(dev) if ds else None
    w = rnd(Cout, 1, 1, C, scale=C ** -0.5).to(dev)
    img = ops.build_weight_planes(w, Cout, 1, C)
    y_ref = ops.bn_add_relu(c3, saved, rows, C, idt=idt, idsaved=idsaved)
    o_ref, p_ref = ops.conv2d_fwd(y_ref, w, N, H, W, C, Cout, 1, 1, 1, 0, None, None, stats=True, wimg=img)
    ops.numerics_status(reset=True)
    o, p, y = ops.conv2d_fwd(c3, w, N, H, W, C, Cout, 1, 1, 1, 0, saved[2], saved[3], stats=True, wimg=img, tail_idt=idt,
                             tail_idsaved=idsaved)
    assert y.dtype == y_ref.dtype and torch.equal(y, y_ref)
    assert torch.equal(o, o_ref) and torch.equal(p, p_ref)
    assert ops.numerics_status() == {"saturated": 0, "nonfinite": 0}
    # against float64: the usual forward bar
    idd = idt.double() * idsaved[2].double() + idsaved[3].double() if ds else idt.double()
    yd = torch.relu(c3.double() * saved[2].double() + saved[3].double() + idd)
    if bf16:
        yd = yd.float().bfloat16().double()
    od = (yd.reshape(rows, C) @ w.reshape(Cout, C).double().t()).reshape(N, H, W, Cout)
    assert rel_err(o.float(), od) < (4e-3 if bf16 else 2e-6)
    # the saturation watch covers the fused loader too
    big = saved.clone()
    big[2] = 5000.0
    ops.conv2d_fwd(c3, w, N, H, W, C, Cout, 1, 1, 1, 0, big[2], big[3], wimg=img, tail_idt=idt, tail_idsaved=idsaved)
    assert ops.numerics_status(reset=True)["saturated"] > 0


def test_losses_on_spatial_logits_with_class_weights_vs_reference(dev):
    """FocalLoss / CrossEntropyLoss as the reference documents them: (b, ch, d0, d1) logits with a (b, d0, d1) target, and
    class weights (_losses.py:36,56-57,91-108) -- fixture F7's extension, produced by the imported reference"""
    from oaprogressionmmf_amd.various import dict_losses
    g = load("f7_focal.npz")
    lg, tg, cw = torch.from_numpy(g["nd:logits"]).to(dev), torch.from_numpy(g["nd:target"]).to(dev), torch.from_numpy(g["nd:class_weight"])
    for tag, w in (("nd", None), ("ndw", cw)):
        for red in ("mean", "sum"):
            x = lg.clone().requires_grad_(True)
            loss = dict_losses["FocalLoss"](reduction=red, gamma=2.0, num_classes=3, class_weight=w)(input=x, target=tg)
            loss.backward()
            ref = float(g[f"{tag}:focal_{red}:loss"])
            assert abs(loss.item() - ref) < 2e-6 * max(1.0, abs(ref)), (tag, red)
            assert rel_err(x.grad, torch.from_numpy(g[f"{tag}:focal_{red}:dlogits"])) < 1e-5, (tag, red)
        x = lg.clone().requires_grad_(True)
        loss = dict_losses["CrossEntropyLoss"](num_classes=3, class_weight=w)(x, tg)
        loss.backward()
        assert abs(loss.item() - float(g[f"{tag}:ce:loss"])) < 2e-6
        assert rel_err(x.grad, torch.from_numpy(g[f"{tag}:ce:dlogits"])) < 1e-5


def test_losses_ignore_index_and_bad_labels(dev):
    """targets outside [0, C): -100 (F.cross_entropy's default ignore_index, which the reference's losses inherit:
    koafusion/various/_losses.py:36,101) gives zero loss and zero gradient; the cross-entropy mean leaves the element out of its
    denominator, the focal loss (reduction 'none', then .mean(): :101-108) still divides by every element.  Any other bad label
    is handled the same way and counted in the numerics status words instead of reading out of bounds."""
    from oaprogressionmmf_amd import ops
    B, C = 37, 3
    lg = rnd(B, C) * 2
    tg = torch.randint(0, C, (B,), generator=G)
    tg[[3, 11, 30]] = -100
    cw = torch.tensor([0.5, 2.0, 1.25])
    for w in (None, cw):
        x = lg.clone().double().requires_grad_(True)
        ce = F.cross_entropy(x, tg, weight=None if w is None else w.double())
        ce.backward()
        loss, dl = ops.focal_loss(lg.to(dev), tg.to(dev), 0.0, focal=False, class_weight=None if w is None else w.to(dev))
        assert abs(loss.item() - ce.item()) < 2e-6 and rel_err(dl.cpu(), x.grad) < 1e-5
        for mean in (True, False):
            x = lg.clone().double().requires_grad_(True)
            logpt = -F.cross_entropy(x, tg, weight=None if w is None else w.double(), reduction="none")
            fl = -((1 - logpt.exp()) ** 2.0) * logpt
            fl = fl.mean() if mean else fl.sum()
            fl.backward()
            loss, dl = ops.focal_loss(lg.to(dev), tg.to(dev), 2.0, mean=mean, class_weight=None if w is None else w.to(dev))
            assert abs(loss.item() - fl.item()) < 2e-6 * max(1.0, abs(fl.item())) and rel_err(dl.cpu(), x.grad) < 1e-5
            assert float(dl[[3, 11, 30]].abs().max()) == 0.0
    ops.numerics_status(reset=True)
    bad = tg.clone()
    bad[5] = 7
    loss, dl = ops.focal_loss(lg.to(dev), bad.to(dev), 2.0)
    assert torch.isfinite(loss).item() and float(dl[5].abs().max()) == 0.0
    assert ops.numerics_status(reset=True)["nonfinite"] == 1


def test_interpolate_any_scale_vs_reference(dev):
    """PTInterpolate for any scale factor (the reference hands the config's `downscale` to F.interpolate: _pt.py:175-192):
    odd sizes, up- and down-scaling, several channels, the (B, CH, D0) linear rank -- fixture F8's extension; the mask branch
    raises ValueError exactly like the reference's (dead) one"""
    from oaprogressionmmf_amd.preproc import PTInterpolate
    g = load("f8_interp.npz")
    mk = lambda name, shape: torch.from_numpy(P.make_input(name, shape)).to(dev)   # noqa: E731
    x, x2, v2, l2 = mk("interp_xr", (2, 1, 70, 50)), mk("interp_xr2", (2, 3, 37, 29)), mk("interp_mr2", (1, 2, 19, 23, 11)), mk("interp_lin", (2, 2, 41))
    for tag, img, sf in (("xr_075", x2, (0.75, 0.75)), ("xr_up", x2, (1.5, 2.0)), ("xr_mix", x, (0.3, 0.85)),
                         ("mr_mix", v2, (0.6, 0.8, 1.0)), ("mr_up", v2, (1.3, 0.5, 2.0)), ("lin", l2, (0.4,))):
        out = PTInterpolate(scale_factor=sf)(img)
        assert tuple(out.shape) == g[tag].shape, (tag, out.shape, g[tag].shape)
        assert float((out.cpu() - torch.from_numpy(g[tag])).abs().max()) < 2e-6 * float(np.abs(g[tag]).max()), tag
    assert str(g["mask_branch"]).startswith("ValueError")
    with pytest.raises(ValueError):
        PTInterpolate(scale_factor=(0.75, 0.6))(x2, (x2 > 0.3).float())
    assert PTInterpolate(scale_factor=(1.0, 1.0))(x2) is x2
    # the recipes' factors still take the average-pooling kernel, with the same values as the general one
    a = PTInterpolate(scale_factor=(0.5, 0.5))(x)
    from oaprogressionmmf_amd import ops
    b = ops.resize(x, [35, 25])
    assert float((a - b).abs().max()) < 1e-6 and rel_err(a, torch.from_numpy(g["xr"])) < 1e-6


def test_adam_amsgrad_and_batchnorm_cumulative_average(dev):
    """the two optimiser / normalisation options the registries accept from torch that were declared holes: Adam(amsgrad=True)
    against torch.optim.Adam on the same gradients, and BatchNorm2d(momentum=None) -- cumulative moving average, factor
    1 / num_batches_tracked -- against nn.BatchNorm2d on the CPU"""
    from oaprogressionmmf_amd import ops
    n = 10007
    p0, gs = rnd(n), [rnd(n) * (3.0 if s == 0 else 1.0) for s in range(4)]       # (a large first gradient: the maximum matters)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=1e-3, weight_decay=1e-4, amsgrad=True)
    pd = p0.clone().to(dev)
    m, v, vmax = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for s in range(4):
        pr.grad = gs[s].clone()
        opt.step()
        ops.adam_step(pd, gs[s].to(dev), m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, s + 1, vmax=vmax)
    assert rel_err(pd, pr.detach()) < 1e-6
    assert rel_err(vmax, opt.state[pr]["max_exp_avg_sq"]) < 1e-6 and float((vmax - v).max()) > 0
    # BatchNorm2d(momentum=None) over three batches
    C, rows = 64, 500
    bn = torch.nn.BatchNorm2d(C, momentum=None).train()
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for it in range(3):
        xb = rnd(rows, C) * (1.0 + it) + 0.3 * it
        bn(xb.t().reshape(1, C, rows, 1))
        ops.bn_finalize(ops.colstats(xb.to(dev), rows, C), C, rows, gam, bet, rm, rv, nbt, -1.0, 1e-5, True)
        assert int(nbt) == it + 1 == int(bn.num_batches_tracked)
        assert rel_err(rm, bn.running_mean) < 1e-5 and rel_err(rv, bn.running_var) < 1e-5


def test_stem_backward_without_the_pool_gradient_and_dc_tensors(dev):
    """stem backward (_torchvision.py:170-174 reversed): the max-pool's input gradient gathered inside the BatchNorm reduction
    (koaf_bn_bwd_reduce_pool) and dc formed by the 7x7 weight gradient on load (koaf_stem_wgrad dy_apply) against the chain of
    separate passes (koaf_maxpool_bwd -> koaf_bn_bwd_reduce -> koaf_bn_bwd_apply -> koaf_stem_wgrad)"""
    from oaprogressionmmf_amd import ops
    N, H, W, C = 3, 70, 58, 64
    x = rnd(N, H, W).to(dev)
    w = rnd(64, 7, 7, 3, scale=0.1).to(dev)
    c0 = ops.stem_fwd(x, ops.stem_fold_w(w), N, H, W)
    H1, W1 = c0.shape[1], c0.shape[2]
    rows = N * H1 * W1
    gam, bet = (rnd(C) * 0.2 + 1).to(dev), (rnd(C) * 0.1).to(dev)
    saved = ops.bn_finalize(ops.colstats(c0, rows, C), C, rows, gam, bet, torch.zeros(C, device=dev), torch.ones(C, device=dev),
                            torch.zeros(1, dtype=torch.int64, device=dev), 0.1, 1e-5, True)
    y, am = ops.maxpool_fwd(c0, saved, N, H1, W1, C)
    dy = rnd(*y.shape).to(dev)
    dg0, db0, dg1, db1 = (torch.empty(C, device=dev) for _ in range(4))
    da0 = ops.maxpool_bwd(dy, am, N, H1, W1, C)
    dc_ref = ops.bn_bwd(da0.clone(), c0, saved, rows, C, rows, dg0, db0, 2, fused=False)
    ap = ops.bn_bwd(None, c0, saved, rows, C, rows, dg1, db1, 2, fused=True, pool=(dy, am, N, H1, W1))
    assert isinstance(ap, ops.BnApply)
    dz_ref = da0 * ((c0 * saved[2] + saved[3]) > 0)
    assert torch.equal(ap.dz, dz_ref)
    assert rel_err(dg1, dg0.double()) < 1e-6 and rel_err(db1, db0.double()) < 1e-6
    assert rel_err(ap.materialize(), dc_ref.double()) < 1e-6
    dw0, dw1 = torch.empty(64, 7, 7, 3, device=dev), torch.empty(64, 7, 7, 3, device=dev)
    ops.stem_wgrad(dc_ref, x, dw0, N, H, W)
    ops.stem_wgrad(ap, x, dw1, N, H, W)
    assert rel_err(dw1, dw0.double()) < 2e-6


@pytest.mark.parametrize("N,H,W,Cin,Cout,bf16", [(3, 24, 24, 256, 64, False), (2, 17, 19, 64, 64, False), (1, 5, 5, 128, 128, False),
                                                 (3, 24, 24, 256, 64, True)])
def test_conv_epilogue_emits_the_consumer_plane_images(dev, N, H, W, Cin, Cout, bf16):
    """KoafEmit: a forward convolution whose OUTPUT BatchNorm is already known (eval mode, rebuilt stages) cuts the plane images
    of relu(sc * y + sh) in its epilogue -- the same bits as koaf_act_planes over the stored y (fp32 and bf16 storage, full and
    ragged last tiles, persistent 1x1 kernels), zero chunk included -- and the following 3x3 convolution picks them up instead
    of running the pre-pass: same output bits."""
    from oaprogressionmmf_amd import ops
    dt = torch.bfloat16 if bf16 else torch.float32
    x = rnd(N, H, W, Cin).to(dev).to(dt)
    w = rnd(Cout, 1, 1, Cin, scale=Cin ** -0.5).to(dev)
    isc, ish = (rnd(Cin) * 0.2 + 1).to(dev), (rnd(Cin) * 0.1).to(dev)
    osc, osh = (rnd(Cout) * 0.3 + 1).to(dev), (rnd(Cout) * 0.2).to(dev)
    img = ops.build_weight_planes(w, Cout, 1, Cin)
    y0, _ = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, isc, ish, wimg=img)
    y1, _ = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, isc, ish, wimg=img, emit=(osc, osh))
    assert torch.equal(y0, y1)
    got = y1._koaf_eplanes[0]
    ref = ops.act_planes(y0, N * H * W, Cout, 1, osc, osh, fscale=ops.ACT_SCALE)
    assert got.shape == ref.shape and torch.equal(got, ref)
    # the consumer: a 3x3 convolution over y1 finds the images (and consumes them), over y0 it cuts its own
    w3 = rnd(Cout, 3, 3, Cout, scale=(9 * Cout) ** -0.5).to(dev)
    img3 = ops.build_weight_planes(w3, Cout, 9, Cout)
    a, _ = ops.conv2d_fwd(y0, w3, N, H, W, Cout, Cout, 3, 3, 1, 1, osc, osh, wimg=img3)
    b, _ = ops.conv2d_fwd(y1, w3, N, H, W, Cout, Cout, 3, 3, 1, 1, osc, osh, wimg=img3, keep_planes=True)
    assert torch.equal(a, b) and not hasattr(y1, "_koaf_eplanes") and b._koaf_xplanes is got
    # the bottleneck-tail form of conv1 (tf 3) emits as well
    if not bf16:
        idt = rnd(N, H, W, Cin).to(dev)
        t0 = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, isc, ish, wimg=img, tail_idt=idt)
        t1 = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, isc, ish, wimg=img, tail_idt=idt, emit=(osc, osh))
        assert torch.equal(t0[0], t1[0]) and torch.equal(t0[2], t1[2])
        assert torch.equal(t1[0]._koaf_eplanes[0], ops.act_planes(t0[0], N * H * W, Cout, 1, osc, osh, fscale=ops.ACT_SCALE))


def _bn_record(ops, dev, x, rows, C):
    return ops.bn_finalize(ops.colstats(x, rows, C), C, rows, (rnd(C) * 0.2 + 1).to(dev), (rnd(C) * 0.1).to(dev), torch.zeros(C, device=dev),
                           torch.ones(C, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), 0.1, 1e-5, True)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(5, 24, 24, 64, 256),       # K = 64: as many k-steps as tiles in flight, several column tiles
                                            (600, 12, 12, 64, 128),     # ... and more tiles than resident blocks: the A stream crosses tile boundaries
                                            (3, 17, 19, 128, 64),       # ragged last tile, 64-column tiles
                                            (2, 16, 16, 256, 512),      # eight k-steps: four k-tiles in flight (one tile per block)
                                            (2, 10, 10, 1024, 256)])
def test_streamed_1x1_kernels_match_the_block_wide_loader(dev, N, H, W, Cin, Cout):
    """koaf_set_stream: the streamed kernel of the dense 1x1 / stride-1 convolutions (every wave loads, transforms and splits its own
    32 rows, k-tiles ahead in registers; KoafGemm A mode M_KS) against the block-wide loader on the same calls: outputs, side-stored
    tails, emitted plane images, per-tile BatchNorm statistics, data gradients with the BatchNorm-backward apply on load and the fused
    reduction -- BIT FOR BIT (same pieces, same MFMA order per accumulator, the statistics as the same tree of 32-row band sums).
    _torchvision.py:118-138 (the 1x1 convolutions of a Bottleneck)."""
    from oaprogressionmmf_amd import ops
    rows = N * H * W
    x = (rnd(N, H, W, Cin) * 1.5 + 0.3).to(dev)
    w = rnd(Cout, 1, 1, Cin, scale=Cin ** -0.5).to(dev)
    img = ops.build_weight_planes(w, Cout, 1, Cin)
    sv = _bn_record(ops, dev, x, rows, Cin)
    shift = (rnd(Cout) * 0.1).to(dev)
    idt = rnd(N, H, W, Cin).to(dev)
    ids = _bn_record(ops, dev, idt, rows, Cin)
    em = ((torch.rand(Cout, generator=G) + 0.5).to(dev), (rnd(Cout) * 0.1).to(dev))

    def forward_calls():
        out = {}
        out["plain"] = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, None, None, stats=True, shift=shift, wimg=img)
        out["prologue"] = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=True, shift=shift, wimg=img)
        out["tail"] = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=True, shift=shift, wimg=img, tail_idt=idt)
        out["tail_ds"] = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=True, shift=shift, wimg=img, tail_idt=idt,
                                        tail_idsaved=ids)
        y, _, yin = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=False, wimg=img, tail_idt=idt, emit=em)
        out["tail_emit"] = (y, None, yin, y._koaf_eplanes[0])
        y, _ = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=False, wimg=img, emit=em)
        out["prologue_emit"] = (y, None, y._koaf_eplanes[0])
        return out

    # the data gradient of a Cout -> Cin convolution contracts over this call's Cin ... reuse the tensors the other way round
    c = x                                                     # conv output whose BatchNorm is back-propagated ([rows, Cin])
    g = (rnd(N, H, W, Cin) * 1e-3).to(dev)
    wd = rnd(Cin, 1, 1, Cout, scale=Cout ** -0.5).to(dev)     # convolution Cout -> Cin
    imgd = ops.build_weight_planes(wd, Cin, 1, Cout)
    dgm, dbt = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev)
    ap = ops.bn_bwd(g.clone(), c, sv, rows, Cin, rows, dgm, dbt, 2, fused=True)
    cx = (rnd(N, H, W, Cout) * 1.5 + 0.3).to(dev)
    savx = _bn_record(ops, dev, cx, rows, Cout)
    res = (rnd(N, H, W, Cout) * 1e-3).to(dev)

    def gradient_calls():
        out = {}
        out["apply"] = (ops.conv2d_dgrad(ap, wd, N, H, W, Cout, Cin, 1, 1, 1, 0, wimg=imgd),)
        out["apply_bnb"] = ops.conv2d_dgrad(ap, wd, N, H, W, Cout, Cin, 1, 1, 1, 0, residual=res, wimg=imgd,
                                            bnb=dict(mode=2, c=cx, saved=savx, dz_amax=True))
        out["apply_bnb_tail"] = ops.conv2d_dgrad(ap, wd, N, H, W, Cout, Cin, 1, 1, 1, 0, residual=res, wimg=imgd,
                                                 bnb=dict(mode=1, c=cx, y=res, saved=savx, c2=cx, saved2=savx, dz_amax=True))
        return out

    was = ops.set_stream(False)
    try:
        f0, g0 = forward_calls(), gradient_calls()
        ops.set_stream(True)
        f1, g1 = forward_calls(), gradient_calls()
    finally:
        ops.set_stream(was)
    for name in f0:
        a, b = f0[name], f1[name]
        assert torch.equal(a[0], b[0]), name                             # the convolution output
        if a[1] is not None:                                              # per-tile statistics: the same tree of band sums in both kernels
            assert torch.equal(a[1], b[1]), name
        for ta, tb in zip(a[2:], b[2:]):                                  # the side-stored tail, the emitted plane images
            assert torch.equal(ta, tb), name
    for name in g0:
        for ta, tb in zip(g0[name], g1[name]):
            assert torch.equal(ta, tb), name


def test_losses_on_segmentation_sized_logits_run_on_a_grid(dev):
    """(b, ch, d0, d1) logits beyond one block's reach (koaf_loss_ws > 0): the focal loss / cross-entropy arithmetic on a grid of
    4096-element blocks with a two-stage fixed-order reduction -- against torch in float64 (class weights, ignored labels), and twice
    for determinism.  _losses.py:36,56-57,91-108."""
    from oaprogressionmmf_amd import ops
    from oaprogressionmmf_amd._lib import lib
    B, C, H, W = 2, 3, 70, 71
    assert lib().koaf_loss_ws(B, H * W) > 0 and lib().koaf_loss_ws(37, 1) == 0
    lg = rnd(B, C, H, W) * 2
    tg = torch.randint(0, C, (B, H, W), generator=G)
    tg[0, 3, 5] = tg[1, 60, 70] = -100
    cw = torch.tensor([0.5, 2.0, 1.25])
    for w in (None, cw):
        wd = None if w is None else w.double()
        x = lg.clone().double().requires_grad_(True)
        ce = F.cross_entropy(x, tg, weight=wd)
        ce.backward()
        loss, dl = ops.focal_loss(lg.to(dev), tg.to(dev), 0.0, focal=False, class_weight=None if w is None else w.to(dev))
        assert abs(loss.item() - ce.item()) < 2e-6 and rel_err(dl.cpu(), x.grad) < 1e-5
        for mean in (True, False):
            x = lg.clone().double().requires_grad_(True)
            logpt = -F.cross_entropy(x, tg, weight=wd, reduction="none")
            fl = -((1 - logpt.exp()) ** 2.0) * logpt
            fl = fl.mean() if mean else fl.sum()
            fl.backward()
            loss, dl = ops.focal_loss(lg.to(dev), tg.to(dev), 2.0, mean=mean, class_weight=None if w is None else w.to(dev))
            assert abs(loss.item() - fl.item()) < 2e-6 * max(1.0, abs(fl.item())) and rel_err(dl.cpu(), x.grad) < 1e-5
            assert float(dl[0, :, 3, 5].abs().max()) == 0.0 and float(dl[1, :, 60, 70].abs().max()) == 0.0
            loss2, dl2 = ops.focal_loss(lg.to(dev), tg.to(dev), 2.0, mean=mean, class_weight=None if w is None else w.to(dev))
            assert loss2.item() == loss.item() and torch.equal(dl2, dl)
