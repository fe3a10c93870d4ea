"""Tensor-level wrappers over the libkoaf C ABI.

Every function takes/returns `torch.Tensor`s that live on a HIP device (torch is used for device
memory and streams only) and launches hand-written gfx950 kernels on torch's current stream.
Activations are NHWC fp32.  Nothing here falls back to torch math: a CPU tensor or a missing
library raises.
"""
import ctypes

import torch

from ._lib import KoafBnApply, KoafBnb, KoafEmit, KoafGemm, KoafError, KoafTail, KoafWImg, check, lib

_i32 = ctypes.c_int32

# Convolutions run on the fp16 contraction scheme (koaf.h: KoafGemm.fmt 1) whenever their operands' scales are known
# (weights: arena plane images; gradients: the amax the BatchNorm backward leaves behind).  KOAF_CONV_FMT=bf16 keeps them
# on the bf16 x 3 scheme for A/B runs.
import os
CONV_F16 = os.environ.get("KOAF_CONV_FMT", "f16") != "bf16"
# activation plane images for the gathered (3x3) convolution kernels (koaf_act_planes); KOAF_APLANES=0: fp32 loaders
APLANES_MASK = int(os.environ.get("KOAF_APLANES", "7"))     # bit 0 forward, 1 data gradient, 2 weight gradient
APLANES = APLANES_MASK != 0
ACT_SCALE = 16.0        # koaf.h KOAF_ACT_SCALE
# Forward plane images of up to this many 16-bit elements are kept on the convolution's output for its weight gradient
# (saves re-cutting them).  Off by default: the mixed lifetimes fragment the caching allocator's pool -- the headline
# step's reserved memory went from 240 to 265 GB (of 288) for 1 % of its time with everything kept, and still to 264 GB
# with only the deep layers' small images.
KEEP_XPLANES_ELEMS = int(os.environ.get("KOAF_KEEP_XPLANES_ELEMS", "0"))

# Optional live profiler (bench.py): when a list is installed here every MFMA-GEMM based call is bracketed
# by two events recorded on the stream the kernel is launched on (torch's current stream) and logged as
# (family, algorithmic_flops, start_event, end_event).
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, family, flops, tag="", elems=0, mpp=6):
    """elems: fp32 elements the call must move through HBM if every operand travels exactly once (its algorithmic bytes / 4);
    mpp: matrix instructions per fp32 product of the call's contraction scheme (6: bf16 x 3, 3: fp16 x 2; koaf.h KoafGemm.fmt)"""
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.append((family, flops, e0, e1, tag, 4.0 * elems, mpp))


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise KoafError("koaf ops need tensors on a HIP device (no CPU fallback exists)")
    if t.dtype not in (torch.float32, torch.int64, torch.uint8, torch.float64, torch.bfloat16):   # float64: reduction workspaces only; bfloat16: activation storage mode
        raise KoafError(f"unexpected dtype {t.dtype}")
    return t.data_ptr()


_STATUS = None       # device int32[4]: the library's numerics status words (koaf.h koaf_set_status_buffer)
_STATUS_DEV = None   # the device they live on: the library holds ONE pointer, the one of the device this process computes on


def _a16(t):
    """the `act16` argument of the entry points (koaf.h): 1 when the call's activation tensor is stored as bf16"""
    return 1 if (t is not None and t.dtype == torch.bfloat16) else 0


def _stream():
    if _STATUS is None or _STATUS_DEV != torch.cuda.current_device():
        _status_buffer()
    return torch.cuda.current_stream().cuda_stream


_STATUS_BY_DEV = {}


def _status_buffer():
    """register the status words every kernel launch may bump: one buffer per device, the library pointing at the CURRENT
    device's (one process = one GPU is the product's layout; a process that moves to another device re-registers there instead
    of letting that device's kernels add into a peer's memory)"""
    global _STATUS, _STATUS_DEV
    d = torch.cuda.current_device()
    if _STATUS is None or _STATUS_DEV != d:
        if d not in _STATUS_BY_DEV:
            _STATUS_BY_DEV[d] = torch.zeros(4, dtype=torch.int32, device=torch.device("cuda", d))
        _STATUS, _STATUS_DEV = _STATUS_BY_DEV[d], d
        check(lib().koaf_set_status_buffer(_STATUS.data_ptr()), "set_status_buffer")
    return _STATUS


def numerics_status(reset=False):
    """{"saturated": ..., "nonfinite": ...} since the last reset (one small device-to-host copy: call it per epoch, not per step).
    saturated: activation elements (or GEMM tiles holding some) that left the fp16 range of the fixed activation scale
    (|x| > 4094) and were clamped, or were not finite; nonfinite: NaN / Inf that reached an operand's scale scalar (the GEMM
    then returned NaN everywhere) or a BatchNorm's coefficients."""
    st = _status_buffer()
    torch.cuda.synchronize(st.device)        # (encoder lanes / side streams add too: every stream of the device has landed)
    v = st.cpu().tolist()
    if reset:
        st.zero_()
    return {"saturated": int(v[0]) & 0xffffffff, "nonfinite": int(v[1]) & 0xffffffff}


def check_numerics(reset=True):
    """numerics_status() + a RuntimeWarning when anything was flagged; returns the status dict"""
    import warnings
    st = numerics_status(reset=reset)
    if st["saturated"]:
        warnings.warn(f"koaf: {st['saturated']} activation elements / tiles exceeded the fixed activation scale's range (|x| > 4094 behind "
                      f"a BatchNorm) and were clamped in the convolution operands: results are no longer at fp32 rounding level",
                      RuntimeWarning, stacklevel=2)
    if st["nonfinite"]:
        warnings.warn(f"koaf: non-finite values reached {st['nonfinite']} operand scales / BatchNorm coefficients (a diverged run)",
                      RuntimeWarning, stacklevel=2)
    return st


def _img(wimg):
    """ctypes view of a weight's plane images: wimg = (F, D, amax) device tensors (F / D may be None) or None"""
    if wimg is None:
        return None
    f, d, amax = wimg
    for t in (f, d):
        if t is not None and (not t.is_cuda or t.dtype != torch.int16):
            raise KoafError("weight plane images are int16 (fp16 bit patterns) tensors on the HIP device")
    return ctypes.byref(KoafWImg(f=f.data_ptr() if f is not None else None, d=d.data_ptr() if d is not None else None,
                                 amax=_ptr(amax)))


class BnApply(object):
    """A gradient w.r.t. a conv output that exists only as its BatchNorm-backward recipe: dc = coef0*dz + coef3 - coef2*c
    (coef [4][C] and the scale bound `amax` from koaf_bn_bwd_finalize).  conv2d_dgrad / conv2d_wgrad take it in place of the
    dy tensor and form dc in their loaders (koaf.h KoafOperand.tf 2), so dc never travels through HBM; materialize()
    writes it out for consumers that are not such GEMMs."""
    __slots__ = ("dz", "c", "coef", "amax", "mean", "rows", "C", "_koaf_planes")

    def __init__(self, dz, c, coef, amax, mean, rows, C):
        self.dz, self.c, self.coef, self.amax, self.mean, self.rows, self.C = dz, c, coef, amax, mean, rows, C
        self._koaf_planes = None        # activation plane images of dc, cut by the first convolution that wants them

    def struct(self):
        return ctypes.byref(KoafBnApply(dz=_ptr(self.dz), c=_ptr(self.c), coef=_ptr(self.coef), amax=_ptr(self.amax)))

    def tensors(self):
        """everything a kernel launched with this recipe reads (to be kept alive / recorded for a second stream)"""
        return (self.dz, self.c, self.coef, self.amax, self._koaf_planes, getattr(self.c, "_koaf_xplanes", None))

    def materialize(self, out=None, want_amax=False):
        dc = out if out is not None else torch.empty(self.c.shape, device=self.c.device, dtype=torch.float32)
        amax = torch.zeros(1, device=self.c.device, dtype=torch.float32) if want_amax else None
        check(lib().koaf_bn_bwd_apply(_ptr(self.dz), _ptr(self.c), _ptr(self.mean), _ptr(self.coef), _ptr(dc), self.rows, self.C,
                                      _ptr(amax), _a16(self.c), _stream()), "bn_bwd_apply")
        if want_amax:
            dc._koaf_amax = amax
        return dc


def _dy_args(dy, dy_amax):
    """(dy pointer, dy_amax pointer, KoafBnApply* or None, reference tensor) of a gradient given as a tensor or as a BnApply"""
    if isinstance(dy, BnApply):
        return None, None, dy.struct(), dy.dz
    return _ptr(dy), _ptr(dy_amax), None, dy


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
def act_planes(x, npix, C, tf=0, sc=None, sh=None, x2=None, sc2=None, amax=None, fscale=0.0):
    """fp16 piece planes [2][npix][C] (+ the zero chunk) of x (tf 0), relu(sc*x+sh) (tf 1) or sc*x + sh - sc2*x2 (tf 2), times
    the operand scale (scale(*amax) or fscale): the A operand of the gathered convolution kernels (koaf.h koaf_act_planes)."""
    L = lib()
    out = torch.empty(L.koaf_act_planes_elems(npix, C), device=x.device, dtype=torch.int16)
    check(L.koaf_act_planes(_ptr(x), _ptr(x2), npix, C, tf, _ptr(sc), _ptr(sh), _ptr(sc2), _ptr(amax), float(fscale),
                            out.data_ptr(), _a16(x2 if tf == 2 else x), _stream()), "act_planes")
    return out


def set_conv3x3_halo(mode):
    """how 3x3 / stride-1 convolutions over plane images run: 0 / False per-tap gather kernel, 1 / True halo kernel with the
    shape picked per layer (default), 2 / 3 the 256- / 128-row halo shape; returns the previous mode (koaf.h
    koaf_set_conv3x3_halo)"""
    return lib().koaf_set_conv3x3_halo(int(mode))


def set_stream(on):
    """the streamed kernel for dense 1x1 / stride-1 convolutions and their data gradients (koaf.h koaf_set_stream): True (default) /
    False = the block-wide loader; returns the previous setting"""
    return bool(lib().koaf_set_stream(1 if on else 0))


def use_aplanes(wimg, KH, KW, C):
    """activation plane images pay where a kernel gathers (every element is otherwise converted KH*KW times)"""
    return APLANES and wimg is not None and KH * KW > 1 and C % 32 == 0


def conv2d_fwd(x, w, N, H, W, Cin, Cout, KH, KW, stride, pad, in_sc=None, in_sh=None, stats=False, shift=None, wimg=None,
               aplanes=None, tail_idt=None, tail_out=None, tail_idsaved=None, keep_planes=False, emit=None):
    """x [N,H,W,Cin] (any view with that memory), w packed [Cout,KH,KW,Cin] -> y [N,OH,OW,Cout],
    (part, rows) per-tile column statistics if stats, summed about `shift` [Cout] (hand the same tensor to bn_finalize).
    wimg = (F, D, amax) plane images of w (arena.weight_planes) or None: with them the contraction runs on the fp16
    scheme (KoafGemm.fmt 1: half the matrix instructions, same accuracy), the weight tiles DMA'd from F.
    tail_idt (1x1 / stride 1 with wimg): x is the PREVIOUS block's raw last conv output c3, in_sc / in_sh its BatchNorm
    coefficients and tail_idt that block's identity: the input y = relu(in_sc*x + in_sh + tail_idt) (the bottleneck tail) is
    formed on load and written to tail_out (allocated here when None); returns (y, part, tail_out).  tail_idsaved: the block had a
    downsample branch -- tail_idt is that branch's raw conv output and tail_idsaved its BatchNorm record (mean, invstd, sc, sh).
    keep_planes: the activation plane images cut for this call stay attached to the output for its weight gradient.
    emit = (sc, sh): the coefficients of the BatchNorm BEHIND this convolution are already known (eval mode, a stage rebuilt in
    backward): the epilogue also cuts the plane images of relu(sc*y + sh) -- bit for bit what the following 3x3 convolution's
    act_planes pre-pass would cut after reading y back -- and leaves them on the output (y._koaf_eplanes), where that convolution
    finds them."""
    L = lib()
    OH, OW = conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad)
    if tail_idt is not None and tail_out is None:
        tail_out = torch.empty_like(x)     # (before the output, like the element-wise tail pass it replaces: same allocation order,
        #                                     same caching-allocator block reuse -- the other order cost 17 GB of reserved memory)
    y = _empty((N, OH, OW, Cout), x, dtype=x.dtype)            # (bf16 activation storage: the output follows the input)
    part, rows = None, _i32(0)
    if stats:
        nrows = L.koaf_conv2d_stats_rows(N * OH * OW, Cout)
        part = _empty((nrows, 2, Cout), x)
    e0 = _prof_begin()
    if aplanes is None:
        aplanes = (APLANES_MASK & 1) and use_aplanes(wimg, KH, KW, Cin) and wimg[0] is not None and stride == 1     # (stride 2: the pre-pass
        #                                           would cut four times the pixels the kernel reads)
    xpl = None
    tail = None
    if tail_idt is not None:
        tail = ctypes.byref(KoafTail(idt=_ptr(tail_idt), y_out=_ptr(tail_out),
                                     idt_sc=_ptr(tail_idsaved[2]) if tail_idsaved is not None else None,
                                     idt_sh=_ptr(tail_idsaved[3]) if tail_idsaved is not None else None))
        aplanes = False
    if torch.is_tensor(aplanes):
        xpl = aplanes                   # images cut by the caller (act_planes with this call's transform and ACT_SCALE)
    elif aplanes:
        ep = getattr(x, "_koaf_eplanes", None)
        if ep is not None:              # (consumed once: the images live on with the convolution that takes them)
            del x._koaf_eplanes
        if (ep is not None and in_sc is not None and ep[1].data_ptr() == in_sc.data_ptr() and ep[2].data_ptr() == in_sh.data_ptr()
                and ep[0].numel() == L.koaf_act_planes_elems(N * H * W, Cin)):      # (the same coefficient memory: rows of one BatchNorm record)
            xpl = ep[0]                 # cut by the producing convolution's epilogue (emit): no pre-pass over x
        else:
            xpl = act_planes(x, N * H * W, Cin, 1 if in_sc is not None else 0, in_sc, in_sh, fscale=ACT_SCALE)
    epl, em = None, None
    if emit is not None:
        epl = torch.empty(L.koaf_act_planes_elems(N * OH * OW, Cout), device=x.device, dtype=torch.int16)
        em = ctypes.byref(KoafEmit(planes=epl.data_ptr(), sc=_ptr(emit[0]), sh=_ptr(emit[1])))
    check(L.koaf_conv2d_fwd(_ptr(x), _ptr(w), _ptr(y), N, H, W, Cin, Cout, KH, KW, stride, pad, _ptr(in_sc),
                            _ptr(in_sh), _ptr(part), ctypes.addressof(rows), _ptr(shift) if stats else None,
                            _img(wimg), xpl.data_ptr() if xpl is not None else None, tail, em, _a16(x), _stream()), "conv2d_fwd")
    if epl is not None:
        y._koaf_eplanes = (epl, emit[0], emit[1])
    if xpl is not None and not torch.is_tensor(aplanes) and (keep_planes or xpl.numel() <= KEEP_XPLANES_ELEMS):
        y._koaf_xplanes = xpl       # ride on the output: this conv's weight gradient reads them instead of cutting them again
    _prof_end(e0, "gemm", 2.0 * N * OH * OW * Cout * KH * KW * Cin,
              f"conv_fwd k{KH}s{stride} {Cin}->{Cout} px{N*OH*OW}" + (" +tail" if tail is not None else ""),
              N * H * W * Cin * (3 if tail is not None else 1) + Cout * KH * KW * Cin + N * OH * OW * Cout * (2 if epl is not None else 1),
              mpp=3 if wimg is not None else 6)
    if stats:
        part = part[:rows.value]
    if tail is not None:
        return y, part, tail_out
    return y, part


def conv2d_dgrad(dy, w, N, H, W, Cin, Cout, KH, KW, stride, pad, residual=None, bnb=None, wimg=None, dy_amax=None,
                 aplanes=None):
    """dx = conv_transpose(dy, w) (+residual).  bnb = dict(mode, c, saved[, y][, c2, saved2]): fuse the
    BatchNorm(+ReLU)-backward reduction of the layer that produced x into the epilogue; then returns
    (dz, part [rows][nsum][Cin]) instead of dx (+ the device scalar max |dz| as a third element when bnb has "dz_amax": True).
    wimg + dy_amax (device scalar max |dy|): fp16 scheme.  dy may be a BnApply (needs wimg with its D image).
    aplanes (None = where it pays: gathered kernels with a D image; stride 2 too -- dy is the SMALL tensor there and its four
    parity-class GEMMs gather from the same images): dy is cut ONCE into activation plane images (the BatchNorm-backward
    apply included) and the kernel's input tiles are DMA'd from them."""
    L = lib()
    dyp, amp, app, like = _dy_args(dy, dy_amax)
    a16 = _a16(dy.c) if isinstance(dy, BnApply) else (_a16(bnb["c"]) if bnb is not None else 0)
    dx = _empty((N, H, W, Cin), like)
    if aplanes is None:
        aplanes = ((APLANES_MASK & 2) and use_aplanes(wimg, KH, KW, Cout) and wimg[1] is not None and stride in (1, 2) and
                   (app is not None or dy_amax is not None))
    e0 = _prof_begin()
    dypl = None
    if aplanes:
        dypl = _dy_planes(dy, dy_amax, N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad), Cout)
    dypp = dypl.data_ptr() if dypl is not None else None
    fl = 2.0 * N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad) * Cout * KH * KW * Cin
    tag = f"conv_dgrad k{KH}s{stride} {Cin}->{Cout} px{N*H*W}"
    el = N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad) * Cout + Cout * KH * KW * Cin + N * H * W * Cin
    el += N * H * W * Cin if residual is not None else 0
    mpp = 3 if (wimg is not None and (dy_amax is not None or app is not None)) else 6
    if app is not None:
        el += N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad) * Cout      # (dz and c are both read)
        tag += " apply"
    if bnb is None:
        check(L.koaf_conv2d_dgrad(dyp, _ptr(w), _ptr(dx), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                  _ptr(residual), _img(wimg), amp, app, dypp, a16, _stream()), "conv2d_dgrad")
        _prof_end(e0, "gemm", fl, tag, el, mpp=mpp)
        return dx
    sv, sv2 = bnb["saved"], bnb.get("saved2")
    dzmax = _empty((1,), like) if bnb.get("dz_amax") else None
    kb = KoafBnb(mode=bnb["mode"], dz_amax=_ptr(dzmax), c=_ptr(bnb["c"]), y=_ptr(bnb.get("y")), sc=_ptr(sv[2]), sh=_ptr(sv[3]),
                 mean=_ptr(sv[0]), invstd=_ptr(sv[1]), c2=_ptr(bnb.get("c2")),
                 mean2=_ptr(sv2[0]) if sv2 is not None else None, invstd2=_ptr(sv2[1]) if sv2 is not None else None)
    nsum = 3 if bnb.get("c2") is not None else 2
    part = _empty((L.koaf_conv2d_dgrad_bnb_rows(N, H, W, Cin, stride), nsum, Cin), like)
    rows = _i32(0)
    check(L.koaf_conv2d_dgrad_bnb(dyp, _ptr(w), _ptr(dx), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                  _ptr(residual), ctypes.byref(kb), _ptr(part), ctypes.addressof(rows), _img(wimg),
                                  amp, app, dypp, a16, _stream()), "conv2d_dgrad_bnb")
    el += N * H * W * Cin * (1 + (bnb.get("y") is not None) + (bnb.get("c2") is not None))
    _prof_end(e0, "gemm", fl, tag + " +bnb", el, mpp=mpp)
    if dzmax is not None:
        return dx, part[:rows.value], dzmax
    return dx, part[:rows.value]


def _dy_planes(dy, dy_amax, npo, Cout):
    """activation plane images of a gradient: a tensor at scale(*dy_amax), or a BnApply with its apply folded in -- cached on
    the BnApply (an immutable recipe), so the data gradient and the weight gradient of one convolution cut them once"""
    if isinstance(dy, BnApply):
        if dy._koaf_planes is None:
            dy._koaf_planes = act_planes(dy.dz, npo, Cout, 2, dy.coef[0], dy.coef[3], x2=dy.c, sc2=dy.coef[2], amax=dy.amax)
        return dy._koaf_planes
    return act_planes(dy, npo, Cout, 0, amax=dy_amax)


def conv2d_wgrad(dy, x, dw, N, H, W, Cin, Cout, KH, KW, stride, pad, in_sc=None, in_sh=None, dy_amax=None, aplanes=None):
    """writes dw (packed [Cout,KH,KW,Cin] memory); dy_amax (device scalar max |dy|): fp16 scheme; dy may be a BnApply.
    aplanes (None = gathered stride-1 kernels on the fp16 scheme): both operands from activation plane images (dy's are
    shared with conv2d_dgrad), moved K-major by LDS-DMA."""
    L = lib()
    dyp, amp, app, like = _dy_args(dy, dy_amax)
    ws = L.koaf_conv2d_wgrad_ws(N, H, W, Cin, Cout, KH, KW, stride, pad)
    slabs = _empty((ws,), like) if ws > 0 else None
    if aplanes is None:
        aplanes = ((APLANES_MASK & 4) and KH * KW > 1 and stride == 1 and Cin % 8 == 0 and Cout % 8 == 0 and
                   (app is not None or dy_amax is not None))
    e0 = _prof_begin()
    dypl = xpl = None
    if aplanes:
        npo = N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad)
        dypl = _dy_planes(dy, dy_amax, npo, Cout)
        # x's images: the ones the forward convolution cut (they ride on its output = the BatchNorm input dy.c), else cut here
        xpl = getattr(dy.c, "_koaf_xplanes", None) if app is not None else None
        if xpl is None or xpl.numel() != L.koaf_act_planes_elems(N * H * W, Cin):
            xpl = act_planes(x, N * H * W, Cin, 1 if in_sc is not None else 0, in_sc, in_sh, fscale=ACT_SCALE)
    check(L.koaf_conv2d_wgrad(dyp, _ptr(x), _ptr(dw), N, H, W, Cin, Cout, KH, KW, stride, pad, _ptr(in_sc),
                              _ptr(in_sh), _ptr(slabs), amp, app, dypl.data_ptr() if dypl is not None else None,
                              xpl.data_ptr() if xpl is not None else None, _a16(x), _stream()), "conv2d_wgrad")
    npx = N * conv_out(H, KH, stride, pad) * conv_out(W, KW, stride, pad)
    _prof_end(e0, "gemm", 2.0 * npx * Cout * KH * KW * Cin,
              f"conv_wgrad k{KH}s{stride} {Cin}->{Cout} px{N*H*W}" + (" apply" if app is not None else ""),
              npx * Cout * (2 if app is not None else 1) + N * H * W * Cin + Cout * KH * KW * Cin,
              mpp=3 if (dy_amax is not None or app is not None) else 6)
    return dw


def gconv_expand_w(w, C, groups):
    """block-diagonal 64-channel slabs of a grouped 3x3 weight; with the convolutions on the fp16 scheme max |w| rides on the result
    (wexp._koaf_amax: the grouped calls then run three MFMAs per product instead of six, koaf.h)"""
    wexp = _empty((C // 64, 64, 9, 64), w)
    amax = torch.zeros(1, device=w.device, dtype=torch.float32) if CONV_F16 else None
    check(lib().koaf_gconv_expand_w(_ptr(w), _ptr(wexp), C, groups, _ptr(amax), _stream()), "gconv_expand_w")
    if amax is not None:
        wexp._koaf_amax = amax
    return wexp


def gconv_compress_dw(dwexp, dw, C, groups):
    check(lib().koaf_gconv_compress_dw(_ptr(dwexp), _ptr(dw), C, groups, _stream()), "gconv_compress_dw")
    return dw


def gconv3x3_fwd(x, wexp, N, H, W, C, stride, in_sc=None, in_sh=None, stats=False, shift=None):
    L = lib()
    OH, OW = conv_out(H, 3, stride, 1), conv_out(W, 3, stride, 1)
    y = _empty((N, OH, OW, C), x, dtype=x.dtype)
    part, rows = None, _i32(0)
    if stats:
        part = _empty(((N * OH * OW + 127) // 128, 2, C), x)
    e0 = _prof_begin()
    wam = getattr(wexp, "_koaf_amax", None)
    check(L.koaf_gconv3x3_fwd(_ptr(x), _ptr(wexp), _ptr(y), N, H, W, C, stride, _ptr(in_sc), _ptr(in_sh), _ptr(part),
                              ctypes.addressof(rows), _ptr(shift) if stats else None, _ptr(wam), _a16(x), _stream()), "gconv3x3_fwd")
    _prof_end(e0, "gemm", 2.0 * N * OH * OW * C * 9 * (C // 32), f"gconv_fwd s{stride} C{C} px{N*OH*OW}",   # algorithmic (32 groups)
              N * H * W * C + N * OH * OW * C + 9 * C * (C // 32), mpp=3 if (wam is not None and not _a16(x)) else 6)
    return y, part


def gconv3x3_dgrad(dy, wexp, N, H, W, C, stride):
    """dy with dy._koaf_amax (max |dy|, left by BnApply.materialize(want_amax=True)) and wexp with its max |w|: fp16 scheme"""
    dx = _empty((N, H, W, C), dy)
    e0 = _prof_begin()
    wam, dam = getattr(wexp, "_koaf_amax", None), getattr(dy, "_koaf_amax", None)
    if wam is None or dam is None:
        wam = dam = None
    check(lib().koaf_gconv3x3_dgrad(_ptr(dy), _ptr(wexp), _ptr(dx), N, H, W, C, stride, _ptr(wam), _ptr(dam), _stream()), "gconv3x3_dgrad")
    _prof_end(e0, "gemm", 2.0 * N * conv_out(H, 3, stride, 1) * conv_out(W, 3, stride, 1) * C * 9 * (C // 32),
              f"gconv_dgrad s{stride} C{C} px{N*H*W}",
              N * H * W * C + N * conv_out(H, 3, stride, 1) * conv_out(W, 3, stride, 1) * C + 9 * C * (C // 32), mpp=3 if wam is not None else 6)
    return dx


def gconv3x3_wgrad(dy, x, N, H, W, C, stride, in_sc=None, in_sh=None):
    L = lib()
    ws = L.koaf_gconv3x3_wgrad_ws(N, H, W, C, stride)
    slabs = _empty((ws,), dy)
    dwexp = _empty((C // 64, 64, 9, 64), dy)
    e0 = _prof_begin()
    dam = getattr(dy, "_koaf_amax", None) if CONV_F16 else None
    check(L.koaf_gconv3x3_wgrad(_ptr(dy), _ptr(x), _ptr(dwexp), N, H, W, C, stride, _ptr(in_sc), _ptr(in_sh),
                                _ptr(slabs), _ptr(dam), _a16(x), _stream()), "gconv3x3_wgrad")
    _prof_end(e0, "gemm", 2.0 * N * conv_out(H, 3, stride, 1) * conv_out(W, 3, stride, 1) * C * 9 * (C // 32),
              f"gconv_wgrad s{stride} C{C} px{N*H*W}",
              N * H * W * C + N * conv_out(H, 3, stride, 1) * conv_out(W, 3, stride, 1) * C + 9 * C * (C // 32),
              mpp=3 if (dam is not None and not _a16(x)) else 6)
    return dwexp


def stem_fold_w(w):
    w1t = _empty((49, 64), w)
    check(lib().koaf_stem_fold_w(_ptr(w), _ptr(w1t), _stream()), "stem_fold_w")
    return w1t


def stem_fwd(x, w1t, N, H, W, dtype=torch.float32, stats=False, shift=None):
    """dtype: storage type of the output activation (torch.float32, or torch.bfloat16: koaf.h "bf16 ACTIVATION STORAGE");
    stats: -> (y, part): the column statistics of y about `shift`, collected by the kernel (what colstats(y) would reduce)"""
    L = lib()
    y = _empty((N, conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3), 64), x, dtype=dtype)
    part = _empty((L.koaf_stem_stats_rows(N, H), 2, 64), x) if stats else None
    check(L.koaf_stem_fwd(_ptr(x), _ptr(w1t), _ptr(y), N, H, W, _ptr(part), _ptr(shift) if stats else None, _a16(y), _stream()), "stem_fwd")
    return (y, part) if stats else y


def stem_wgrad(dy, x, dw, N, H, W):
    """writes dw (packed [64,7,7,3] memory); dy may be a BnApply (the stem BatchNorm's backward, formed on load)"""
    L = lib()
    dyp, _, app, like = _dy_args(dy, None)
    slabs = _empty((L.koaf_stem_wgrad_ws(N, H, W),), like)
    dw1t = _empty((49, 64), like)
    check(L.koaf_stem_wgrad(dyp, _ptr(x), _ptr(dw1t), N, H, W, _ptr(slabs), app, _a16(dy.c) if app is not None else 0, _stream()),
          "stem_wgrad")
    check(L.koaf_stem_unfold_dw(_ptr(dw1t), _ptr(dw), _stream()), "stem_unfold_dw")
    return dw


# ------------------------------------------------------------------------------------------------
# batch norm
# ------------------------------------------------------------------------------------------------
def colstats(x, rows, C, shift=None):
    L = lib()
    part = _empty((L.koaf_colpart_rows(rows, C), 2, C), x)      # (fp32 whatever the storage type of x)
    r = _i32(0)
    check(L.koaf_colstats(_ptr(x), rows, C, _ptr(part), ctypes.addressof(r), _ptr(shift), _a16(x), _stream()), "colstats")
    return part


def _reduce_ws(rows, C, like):
    """fp64 workspace of the two-stage partial-row reduction (None when one stage does)"""
    n = lib().koaf_bn_reduce_ws(rows, C)
    return torch.empty(n // 8, dtype=torch.float64, device=like.device) if n else None


def bn_finalize(part, C, count, gamma, beta, running_mean, running_var, nbt, momentum, eps, train, shift=None):
    """-> saved [4][C] = mean, invstd, sc, sh.  shift = the tensor the statistics in `part` were summed about (it may be
    running_mean itself: the kernel reads it before updating)"""
    saved = _empty((4, C), gamma if gamma is not None else running_mean)
    rows = part.shape[0] if part is not None else 0
    ws = _reduce_ws(rows, C, saved) if train else None
    check(lib().koaf_bn_finalize(_ptr(part), rows, C, count, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                 _ptr(running_var), _ptr(nbt), momentum, eps, 1 if train else 0, _ptr(saved[0]),
                                 _ptr(saved[1]), _ptr(saved[2]), _ptr(saved[3]), _ptr(shift) if train else None, _ptr(ws),
                                 _stream()), "bn_finalize")
    return saved


def bn_add_relu(c, saved, rows, C, idt=None, idsaved=None, out=None):
    y = out if out is not None else torch.empty_like(c)
    check(lib().koaf_bn_add_relu(_ptr(c), _ptr(saved[2]), _ptr(saved[3]), _ptr(idt),
                                 _ptr(idsaved[2]) if idsaved is not None else None,
                                 _ptr(idsaved[3]) if idsaved is not None else None, _ptr(y), rows, C, _a16(c), _stream()),
          "bn_add_relu")
    return y


def bn_bwd(g, c, saved, rows, C, count, dgamma, dbeta, mask_mode, ymask=None, dz_out=None, dc_out=None, fused=False, pool=None):
    """Full BatchNorm(+ReLU mask) backward: reduce -> finalize -> dc.  g is the upstream gradient; with a mask the masked
    gradient dz is written to dz_out (default: in place over g).  fused=False: dc is written out (koaf_bn_bwd_apply) and
    returned; fused=True: returns a BnApply -- the recipe of dc for the GEMM loaders -- and nothing is written."""
    L = lib()
    if pool is not None:
        # g is the gradient of the max-pool behind this BatchNorm(+ReLU): pool = (pool_g [N,OH,OW,C], argmax, N, H, W); the pool's
        # input gradient is gathered by the reduction itself (koaf_bn_bwd_reduce_pool) and only the masked dz is written
        pg, am, pN, pH, pW = pool
        assert g is None and mask_mode == 2 and rows == pN * pH * pW
        dz = dz_out if dz_out is not None else _empty((pN, pH, pW, C), pg)
        part = _empty((L.koaf_colpart_rows(rows, C), 2, C), pg)
        r = _i32(0)
        dzmax = _empty((1,), pg) if fused else None
        check(L.koaf_bn_bwd_reduce_pool(_ptr(pg), _ptr(am), _ptr(c), _ptr(saved[2]), _ptr(saved[3]), _ptr(saved[0]), _ptr(saved[1]),
                                        _ptr(dz), _ptr(part), ctypes.addressof(r), pN, pH, pW, C, _ptr(dzmax), _a16(c), _stream()),
              "bn_bwd_reduce_pool")
        return _bn_bwd_tail(part[:r.value], 2, 1, dz, c, saved, rows, C, count, dgamma, dbeta, dc_out, fused, dzmax)
    if mask_mode != 0 and dz_out is None:
        dz_out = g  # mask in place
    part = _empty((L.koaf_colpart_rows(rows, C), 2, C), g)
    r = _i32(0)
    dzmax = _empty((1,), g) if fused else None
    check(L.koaf_bn_bwd_reduce(_ptr(g), _ptr(c), _ptr(ymask), _ptr(saved[2]), _ptr(saved[3]), _ptr(saved[0]),
                               _ptr(saved[1]), mask_mode, _ptr(dz_out), _ptr(part), ctypes.addressof(r), rows, C,
                               _ptr(dzmax), _a16(c), _stream()), "bn_bwd_reduce")
    dz = dz_out if dz_out is not None else g
    return _bn_bwd_tail(part[:r.value], 2, 1, dz, c, saved, rows, C, count, dgamma, dbeta, dc_out, fused, dzmax)


def _bn_bwd_tail(part, nsum, i1, dz, c, saved, rows, C, count, dgamma, dbeta, dc_out, fused, dzmax):
    L = lib()
    coef = _empty((4 if fused else 3, C), dz)
    amax = _empty((1,), dz) if fused else None
    check(L.koaf_bn_bwd_finalize(_ptr(part), part.shape[0], C, count, _ptr(saved[2]), _ptr(saved[1]), _ptr(dgamma),
                                 _ptr(dbeta), _ptr(coef), nsum, i1, _ptr(_reduce_ws(part.shape[0], C, dz)),
                                 _ptr(saved[0]) if fused else None, _ptr(dzmax), _ptr(amax), _stream()), "bn_bwd_finalize")
    if fused:
        return BnApply(dz, c, coef, amax, saved[0], rows, C)
    dc = dc_out if dc_out is not None else torch.empty(c.shape, device=c.device, dtype=torch.float32)
    check(L.koaf_bn_bwd_apply(_ptr(dz), _ptr(c), _ptr(saved[0]), _ptr(coef), _ptr(dc), rows, C, None, _a16(c), _stream()),
          "bn_bwd_apply")
    return dc


def bn_bwd_from_part(part, nsum, i1, dz, c, saved, rows, C, count, dgamma, dbeta, dc_out=None, fused=False, dzmax=None):
    """BatchNorm backward when the reduction already happened in a dgrad epilogue: finalize (+ apply unless fused; fused
    needs dzmax, the device scalar max |dz| that epilogue left)."""
    if fused and dzmax is None:
        raise KoafError("bn_bwd_from_part(fused=True) needs dzmax (conv2d_dgrad with bnb['dz_amax'])")
    return _bn_bwd_tail(part, nsum, i1, dz, c, saved, rows, C, count, dgamma, dbeta, dc_out, fused, dzmax)


def maxpool_fwd(c, saved, N, H, W, C):
    OH, OW = conv_out(H, 3, 2, 1), conv_out(W, 3, 2, 1)
    y = _empty((N, OH, OW, C), c, dtype=c.dtype)
    am = _empty((N, OH, OW, C), c, dtype=torch.uint8)
    check(lib().koaf_maxpool_fwd(_ptr(c), _ptr(saved[2]), _ptr(saved[3]), _ptr(y), _ptr(am), N, H, W, C, _a16(c), _stream()),
          "maxpool_fwd")
    return y, am


def maxpool_bwd(dy, am, N, H, W, C):
    da = _empty((N, H, W, C), dy)
    check(lib().koaf_maxpool_bwd(_ptr(dy), _ptr(am), _ptr(da), N, H, W, C, _stream()), "maxpool_bwd")
    return da


def gap_fwd(y, N, HW, C):
    out = _empty((N, C), y)
    check(lib().koaf_gap_fwd(_ptr(y), _ptr(out), N, HW, C, _a16(y), _stream()), "gap_fwd")
    return out


def gap_bwd(dout, N, HW, C):
    dy = _empty((N, HW, C), dout)
    check(lib().koaf_gap_bwd(_ptr(dout), _ptr(dy), N, HW, C, _stream()), "gap_bwd")
    return dy


# ------------------------------------------------------------------------------------------------
# input plumbing
# ------------------------------------------------------------------------------------------------
def slice_fold(x, B, R, Cc, S):
    out = _empty((B * S, R, Cc), x)
    check(lib().koaf_slice_fold(_ptr(x), _ptr(out), B, R, Cc, S, _stream()), "slice_fold")
    return out


def resize(x, out_size):
    """F.interpolate(x, size given as recompute_scale_factor computes it, linear | bilinear | trilinear, align_corners=False) of
    a contiguous fp32 (B, CH, d0[, d1[, d2]]) tensor -> (B, CH, *out_size)  (koaf_resize)"""
    nd = x.dim() - 2
    if nd < 1 or nd > 3 or len(out_size) != nd:
        raise KoafError(f"resize: (B, CH, 1..3 spatial dims) tensors, got {tuple(x.shape)} -> {tuple(out_size)}")
    out = _empty(tuple(x.shape[:2]) + tuple(int(v) for v in out_size), x)
    ins = (ctypes.c_int32 * nd)(*[int(v) for v in x.shape[2:]])
    outs = (ctypes.c_int32 * nd)(*[int(v) for v in out_size])
    check(lib().koaf_resize(_ptr(x), _ptr(out), int(x.shape[0]) * int(x.shape[1]), nd, ctypes.addressof(ins), ctypes.addressof(outs),
                            _stream()), "resize")
    return out


def downscale2(x, B, R, Cc, S, fs):
    out = _empty((B, R // 2, Cc // 2, S // fs), x)
    check(lib().koaf_downscale2(_ptr(x), _ptr(out), B, R, Cc, S, fs, _stream()), "downscale2")
    return out


# ------------------------------------------------------------------------------------------------
# transformer pieces
# ------------------------------------------------------------------------------------------------
_WIDEN = {torch.uint8: 1, torch.uint16: 2, torch.int16: 3}


def widen(x):
    """uint8 / uint16 / int16 device tensor -> fp32 (koaf_widen); fp32 passes through"""
    if x.dtype == torch.float32:
        return x
    if x.dtype not in _WIDEN or not x.is_cuda:
        raise KoafError(f"widen: uint8 / uint16 / int16 tensors on the HIP device, got {x.dtype} on {x.device}")
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib().koaf_widen(x.data_ptr(), _WIDEN[x.dtype], _ptr(y), x.numel(), _stream()), "widen")
    return y


def minmax(x, B):
    """per-sample (min, max) of a contiguous batch -> [B, 2]"""
    n = x.numel() // B
    mm = _empty((B, 2), x)
    ws = _empty((B * lib().koaf_minmax_ws(n),), x)
    check(lib().koaf_minmax(_ptr(x), B, n, _ptr(mm), _ptr(ws), _stream()), "minmax")
    return mm


def augment(x, mm, params, B, R, C, S, mean, std):
    y = torch.empty_like(x)
    check(lib().koaf_augment(_ptr(x), _ptr(y), _ptr(mm), _ptr(params), B, R, C, S, mean, std, _stream()), "augment")
    return y


def linear_fwd(x, w, b, M, N, K, residual=None):
    L = lib()
    y = _empty((M, N), x)
    n = L.koaf_linear_ws(M, N, K)
    ws = _empty((n,), x) if n > 0 else None
    e0 = _prof_begin()
    check(L.koaf_linear_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(residual), _ptr(y), _ptr(ws), M, N, K, _stream()),
          "linear_fwd")
    _prof_end(e0, "gemm", 2.0 * M * N * K, f"linear_fwd M{M} N{N} K{K}",
              M * K + N * K + M * N * (2 if residual is not None else 1))
    return y


def linear_dgrad(dy, w, M, N, K, residual=None):
    L = lib()
    dx = _empty((M, K), dy)
    n = L.koaf_linear_ws(M, K, N)
    ws = _empty((n,), dy) if n > 0 else None
    e0 = _prof_begin()
    check(L.koaf_linear_dgrad(_ptr(dy), _ptr(w), _ptr(residual), _ptr(dx), _ptr(ws), M, N, K, _stream()),
          "linear_dgrad")
    _prof_end(e0, "gemm", 2.0 * M * N * K, f"linear_dgrad M{M} N{N} K{K}",
              M * N + N * K + M * K * (2 if residual is not None else 1))
    return dx


def linear_wgrad(dy, x, dw, db, M, N, K):
    L = lib()
    ws = None
    if db is not None:
        n = L.koaf_colsum_ws(M, N)
        ws = _empty((n,), dy) if n > 0 else None
    e0 = _prof_begin()
    check(L.koaf_linear_wgrad(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), _ptr(ws), M, N, K, _stream()), "linear_wgrad")
    _prof_end(e0, "gemm", 2.0 * M * N * K, f"linear_wgrad M{M} N{N} K{K}", M * N + M * K + N * K)


def layernorm_fwd(x, gamma, beta, rows, D, eps):
    y = torch.empty_like(x)
    mean = _empty((rows,), x)
    rstd = _empty((rows,), x)
    check(lib().koaf_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), rows, D, eps,
                                   _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, rows, D):
    L = lib()
    dx = torch.empty_like(x)
    part = _empty((L.koaf_layernorm_bwd_ws(rows, D),), x)
    check(L.koaf_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dgamma),
                               _ptr(dbeta), _ptr(part), rows, D, _stream()), "layernorm_bwd")
    return dx


def attention_fwd(qkv, B, n, h, d, scale):
    attn = _empty((B, h, n, n), qkv)
    out = _empty((B, n, h * d), qkv)
    e0 = _prof_begin()
    check(lib().koaf_attention_fwd(_ptr(qkv), _ptr(attn), _ptr(out), B, n, h, d, scale, _stream()), "attention_fwd")
    _prof_end(e0, "gemm", 4.0 * B * h * n * n * d, f"attn_fwd B{B} n{n}", B * n * h * d * 4 + B * h * n * n)
    return out, attn


def attention_bwd(dout, qkv, attn, B, n, h, d, scale):
    dqkv = torch.empty_like(qkv)
    ws = torch.empty_like(attn)
    e0 = _prof_begin()
    check(lib().koaf_attention_bwd(_ptr(dout), _ptr(qkv), _ptr(attn), _ptr(dqkv), _ptr(ws), B, n, h, d, scale,
                                   _stream()), "attention_bwd")
    _prof_end(e0, "gemm", 8.0 * B * h * n * n * d, f"attn_bwd B{B} n{n}", B * n * h * d * 7 + B * h * n * n)
    return dqkv


def gelu_fwd(x):
    y = torch.empty_like(x)
    check(lib().koaf_gelu_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), "gelu_fwd")
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    check(lib().koaf_gelu_bwd(_ptr(dy), _ptr(x), _ptr(dx), x.numel(), _stream()), "gelu_bwd")
    return dx


def relu_fwd(x):
    y = torch.empty_like(x)
    check(lib().koaf_relu_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), "relu_fwd")
    return y


def relu_bwd(dy, y):
    dx = torch.empty_like(y)
    check(lib().koaf_relu_bwd(_ptr(dy), _ptr(y), _ptr(dx), y.numel(), _stream()), "relu_bwd")
    return dx


def dropout(x, p, seed, epoch=None):
    """epoch: optional int64 device scalar folded into the seed (functional.DeviceStepState)"""
    y = torch.empty_like(x)
    check(lib().koaf_dropout(_ptr(x), _ptr(y), x.numel(), p, seed, _ptr(epoch), _stream()), "dropout")
    return y


def dropout2d(x, N, HW, C, p, seed, epoch=None):
    y = torch.empty_like(x)
    check(lib().koaf_dropout2d(_ptr(x), _ptr(y), N, HW, C, p, seed, _ptr(epoch), _stream()), "dropout2d")
    return y


def counter_add(counter, delta=1):
    """counter (int64 device scalar) += delta, on the stream (a graph node when the step is captured)"""
    check(lib().koaf_counter_add(_ptr(counter), delta, _stream()), "counter_add")


def add(a, b):
    out = torch.empty_like(a)
    check(lib().koaf_add(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "add")
    return out


def focal_loss(logits, target, gamma, mean=True, focal=True, class_weight=None):
    """logits (B, C[, d0, d1, ...]) contiguous, target (B[, d0, d1, ...]) int64, class_weight (C,) or None -> (loss, dlogits)"""
    B, C = logits.shape[0], logits.shape[1]
    S = 1
    for d in logits.shape[2:]:
        S *= int(d)
    if tuple(target.shape) != (B,) + tuple(logits.shape[2:]):
        raise KoafError(f"loss: target shape {tuple(target.shape)} does not match logits {tuple(logits.shape)}")
    loss = _empty((), logits)
    dl = torch.empty_like(logits)
    nws = lib().koaf_loss_ws(B, S)          # (segmentation-sized inputs: the grid form's partial sums)
    ws = _empty((nws,), logits) if nws > 0 else None
    if focal:
        check(lib().koaf_focal_loss(_ptr(logits), _ptr(target), _ptr(class_weight), _ptr(loss), _ptr(dl), B, C, S, gamma,
                                    1 if mean else 0, _ptr(ws), _stream()), "focal_loss")
    else:
        check(lib().koaf_ce_loss(_ptr(logits), _ptr(target), _ptr(class_weight), _ptr(loss), _ptr(dl), B, C, S, _ptr(ws), _stream()), "ce_loss")
    return loss, dl


def adam_step(p, g, m, v, n, lr, b1, b2, eps, wd, step, adamw=False, hyper=None, vmax=None):
    """hyper: optional device float[3] from adam_hyper() -- lr and step then come from the device (captured steps);
    vmax: amsgrad's running maximum of the second moment (updated in place)"""
    check(lib().koaf_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, b1, b2, eps, wd, step, 1 if adamw else 0,
                               _ptr(hyper), _ptr(vmax), _stream()), "adam_step")


def adam_hyper(step, lr, b1, b2, hyper):
    """++step (int32 device scalar); hyper[3] = {lr, lr / (1 - b1^step), sqrt(1 - b2^step)} from the device scalars"""
    if step.dtype != torch.int32 or not step.is_cuda:
        raise KoafError("adam_hyper: step is an int32 device scalar")
    check(lib().koaf_adam_hyper(step.data_ptr(), _ptr(lr), b1, b2, _ptr(hyper), _stream()), "adam_hyper")


def build_weight_planes(w, R, taps, C):
    """(F, D, amax) fp16 plane images (int16 tensors) + device scalar max |w| of ONE weight w [R][taps][C] (packed conv
    weight): what arena.ParamArena keeps for every convolution weight of a model, for callers without an arena (tests,
    micro-benchmarks)."""
    from ._lib import KoafWPlane
    Kp, Rp = (taps * C + 31) // 32 * 32, (R + 31) // 32 * 32
    nf, nd = 2 * R * Kp, 2 * C * taps * Rp
    d_off = (nf + 63) // 64 * 64
    planes = torch.zeros(d_off + nd, device=w.device, dtype=torch.int16)
    amax = torch.zeros(1, device=w.device, dtype=torch.float32)
    ent = KoafWPlane(src_off=0, f_off=0, d_off=d_off, tile0=0, R=R, taps=taps, C=C, Kp=Kp, Rp=Rp)
    tab = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).to(w.device)
    ntiles = ((R + 31) // 32) * taps * ((C + 31) // 32)
    check(lib().koaf_wplanes_build(_ptr(w), planes.data_ptr(), _ptr(amax), tab.data_ptr(), 1, ntiles, _stream()),
          "wplanes_build")
    return planes[:nf], planes[d_off:d_off + nd], amax


def gemm(desc: KoafGemm):
    check(lib().koaf_gemm(ctypes.byref(desc), _stream()), "gemm")
