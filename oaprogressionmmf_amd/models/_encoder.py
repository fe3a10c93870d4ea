"""The slice-wise 2-D CNN encoder as ONE fused HIP schedule (forward and backward).

Reference: the `nn.Sequential(*list(resnet.children())[:-1])` trunks of
koafusion/models/_xrNmrMcP.py:47-59 / _xr1_cnn.py:17-21 executing koafusion/models/_torchvision.py
(stem :170-174, Bottleneck.forward :118-138, BasicBlock.forward :62-80, avgpool :182).

MI355X-first layout of the computation (all activations NHWC fp32, rows = (image, pixel)):
  * every conv is an implicit GEMM on the MFMA kernel (koaf_gemm.hip: fp32 in/out, split-bf16 products); its epilogue emits the
    per-channel partial sums the following train-mode BatchNorm needs;
  * BatchNorm+ReLU are never materialised: the consumer conv applies relu(sc*x+sh) while it loads its
    operand (forward A operand, wgrad B operand), so each conv output is written once and only the raw
    conv outputs are kept for backward;
  * the bottleneck tail relu(bn3(c3) + identity) is one elementwise kernel; backward runs the
    BatchNorm reductions / applies as HBM-bound kernels between the dgrad / wgrad GEMMs, and the
    identity gradient is added in the conv1-dgrad epilogue;
  * the 3 identical input channels of `repeat(..., k=3)` are folded into the stem weights;
  * ResNeXt's grouped 3x3 runs on the same GEMM as 64-channel block-diagonal slabs.
Parameter gradients are written straight into the flat gradient arena.
"""
import torch
from torch import nn

from .. import ops
from ..arena import deliver_grad, grad_target, packed_weight, weight_planes
from ._core_fes import BasicBlock, Bottleneck


import os

USE_SIDE_STREAM = os.environ.get("KOAF_SIDE_STREAM", "1") != "0"


class _Rec:
    __slots__ = ("kind", "blk", "yin", "c1", "s1", "c2", "s2", "c3", "s3", "cd", "sd", "y", "dims", "wexp", "tail")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)


def _stat_shift(bn, train):
    """statistics of a train-mode BatchNorm input are summed about its running mean (see KoafGemm.stats_shift)"""
    return bn.running_mean if (train and bn is not None and bn.running_mean is not None) else None


# The bottleneck tail y = relu(bn3(c3) + identity) of a block (identity = the block input, or BatchNorm(downsample conv)) is not
# run as an element-wise pass: the NEXT block's conv1 forms it while it loads its operand and writes y once (ops.conv2d_fwd tail_idt; koaf.h
# KoafOperand.tf 3) -- 12 B per element of c3 / identity / y traffic instead of 12 + 4.  KOAF_FUSE_TAIL=0 keeps the pass.
FUSE_TAIL = os.environ.get("KOAF_FUSE_TAIL", "1") != "0"
FUSE_TAIL_DS = os.environ.get("KOAF_FUSE_TAIL_DS", "1") != "0"      # ... also behind blocks with a downsample branch


def _can_take_tail(blk):
    """can this block's first convolution form the previous block's tail on load?  (1x1 / stride 1 on the fp16 scheme with
    its weight plane images current)"""
    c = getattr(blk, "conv1", None)
    if not (FUSE_TAIL and isinstance(blk, Bottleneck) and c is not None and ops.CONV_F16):
        return False
    if c.kernel_size != (1, 1) or c.stride != (1, 1) or c.groups != 1 or c.in_channels % 32:
        return False
    img = weight_planes(c.weight)
    return img is not None and img[0] is not None


KEEP_PLANES_RECOMPUTE = os.environ.get("KOAF_KEEP_PLANES_RECOMPUTE", "1") != "0"


# The plane images a 3x3 convolution gathers from are cut by the epilogue of the convolution that PRODUCES its input whenever the
# BatchNorm between the two is already known -- eval mode, and every stage rebuilt in backward from its saved statistics -- instead
# of by a pass of their own over the stored tensor (ops.conv2d_fwd emit; koaf.h KoafEmit).  KOAF_EMIT_PLANES=0 keeps the pass.
EMIT_PLANES = os.environ.get("KOAF_EMIT_PLANES", "1") != "0"


def _takes_planes(conv):
    """does this convolution gather its input from activation plane images (ops.conv2d_fwd's own rule)?"""
    if not (EMIT_PLANES and ops.CONV_F16 and (ops.APLANES_MASK & 1)):
        return False
    if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.groups != 1:
        return False
    img = weight_planes(conv.weight)
    return img is not None and img[0] is not None and ops.use_aplanes(img, 3, 3, conv.in_channels)


def _conv_fwd(x, conv, N, H, W, in_saved, train, bn=None, tail_idt=None, tail_out=None, tail_idsaved=None, keep_planes=False, emit=None):
    """bn: the BatchNorm that consumes this conv's output statistics; tail_idt: x / in_saved are the previous block's last
    conv output and BatchNorm, tail_idt its identity -- the input is their bottleneck tail, formed on load (then the sixth
    return value is that input, written by the convolution)"""
    w = packed_weight(conv.weight)
    cin, cout = conv.in_channels, conv.out_channels
    k, s, p, g = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.groups
    sc, sh = (in_saved[2], in_saved[3]) if in_saved is not None else (None, None)
    wexp = None
    shift = _stat_shift(bn, train)
    if tail_idt is not None:
        y, part, yin = ops.conv2d_fwd(x, w, N, H, W, cin, cout, k, k, s, p, sc, sh, stats=train, shift=shift,
                                      wimg=weight_planes(conv.weight), tail_idt=tail_idt, tail_out=tail_out,
                                      tail_idsaved=tail_idsaved, emit=emit)
        return y, part, ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p), wexp, yin
    if g == 1:
        y, part = ops.conv2d_fwd(x, w, N, H, W, cin, cout, k, k, s, p, sc, sh, stats=train, shift=shift,
                                 wimg=weight_planes(conv.weight), keep_planes=keep_planes, emit=emit)
    else:
        if k != 3 or p != 1 or cin != cout:
            raise NotImplementedError("grouped convolution other than the ResNeXt 3x3 is not built")
        wexp = ops.gconv_expand_w(w, cin, g)
        y, part = ops.gconv3x3_fwd(x, wexp, N, H, W, cin, s, sc, sh, stats=train, shift=shift)
    return y, part, ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p), wexp


def _bn_fin(bn, part, count):
    train = bn.training or bn.running_mean is None
    # momentum None = cumulative moving average (factor 1 / num_batches_tracked): -1 to the kernel (koaf.h koaf_bn_finalize)
    return ops.bn_finalize(part if train else None, bn.num_features, count, bn.weight.detach(), bn.bias.detach(),
                           bn.running_mean, bn.running_var, bn.num_batches_tracked if train else None,
                           -1.0 if bn.momentum is None else bn.momentum, bn.eps, train, shift=_stat_shift(bn, train))


_LANES = {}


def lane_streams(device, lane):
    """(main, side) HIP streams of encoder lane `lane` on `device` (created once)."""
    key = (device.index, lane)
    if key not in _LANES:
        # critical path (forward chain, dgrad -> BatchNorm backward) on a HIGH priority stream, the weight
        # gradients on a LOW priority one: wgrad blocks are dispatched only into what the critical path leaves
        # free, so they trail behind and fill the HBM-bound BatchNorm phases and GEMM tails
        hi, lo = _priorities()
        _LANES[key] = (torch.cuda.Stream(device=device, priority=hi), torch.cuda.Stream(device=device, priority=lo))
    return _LANES[key]


def _priorities():
    try:
        lo, hi = torch.cuda.Stream.priority_range()   # (least, greatest); numerically lower = higher priority
    except Exception:  # noqa: BLE001
        lo, hi = 0, -1
    if os.environ.get("KOAF_STREAM_PRIORITY", "0") == "0":   # measured: 2 levels only; prioritising hurts (163 vs 145 ms)
        return 0, 0
    return hi, lo


_JOIN = {"queued": False, "streams": []}
CALLER_STREAM = None     # set by run_trunks around the lane launches: the stream the model's forward was called on


def _final_join():
    """End of the backward pass: the stream the step runs on waits for every encoder lane that ran.  That stream is the
    one the forward was called on (recorded then): this callback may run on autograd's worker thread, whose own current
    stream is not the caller's -- under HIP-graph capture it is not even part of the capture."""
    for st, caller in _JOIN["streams"]:
        (caller if caller is not None else torch.cuda.current_stream()).wait_stream(st)
    _JOIN["streams"] = []
    _JOIN["queued"] = False


def _register_join(stream, caller=None):
    _JOIN["streams"].append((stream, caller))
    if not _JOIN["queued"]:
        _JOIN["queued"] = True
        from torch.autograd import Variable
        Variable._execution_engine.queue_callback(_final_join)


class _SideStream:
    """Second HIP stream for the weight gradients.  In backward every wgrad GEMM is off the critical path
    (dgrad -> BatchNorm backward -> next dgrad), so they run here: MFMA-bound wgrad blocks fill the CUs while the
    main stream runs HBM-bound BatchNorm kernels or the tail of a dgrad.  Gradients are delivered (p.grad /
    data-parallel hook) only after the main stream has joined this one.

    Two ways to keep the tensors a weight gradient reads alive under the lagging stream: record_stream() (free running:
    one join at the end of the encoder's backward), or -- `hold`, for activation recompute -- this object keeps the
    references and the main stream joins after every block: tensors handed over with record_stream() are recycled by the
    caching allocator only once the side stream's events have completed, which at recompute-sized footprints (a 102 GB stage
    rebuilt at a time) drove the reserved pool to the HBM limit and every allocation into a synchronising retry; with a join
    per block the main stream is ordered behind the reads before the tensors die, so plain stream-ordered reuse is safe."""
    _streams = {}

    def __init__(self, device, stream=None, hold=False):
        if stream is None:
            key = (device.type, device.index)
            if key not in _SideStream._streams:
                _SideStream._streams[key] = torch.cuda.Stream(device=device, priority=_priorities()[1])
            stream = _SideStream._streams[key]
        self.stream = stream
        self.pending = []
        self.hold = hold
        self.held = []

    def run(self, inputs, fn, param, buf, acc):
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            fn()
        if self.hold:
            self.held.append(inputs)
        else:
            for t in inputs:
                if t is not None:
                    t.record_stream(self.stream)     # the caching allocator must not recycle them under the side stream
        self.pending.append((param, buf, acc))

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        for param, buf, acc in self.pending:
            deliver_grad(param, buf, acc)
        self.pending = []
        self.held = []


# The BatchNorm-backward "apply" (dc = coef0*dz + coef3 - coef2*c) is formed in the loaders of the dgrad / wgrad GEMMs that
# consume dc (ops.BnApply, koaf.h KoafOperand.tf 2) instead of being written out by an element-wise pass.  KOAF_FUSE_APPLY=0
# (or convolutions off the fp16 scheme) materialises dc as before.
FUSE_APPLY = os.environ.get("KOAF_FUSE_APPLY", "1") != "0"
WGRAD_EARLY = os.environ.get("KOAF_WGRAD_EARLY", "0") == "1"


def _conv_bwd(conv, dc, x, N, H, W, in_saved, wexp, residual=None, need_dx=True, side=None, bnb=None):
    """weight gradient (x transformed on load by in_saved) and data gradient of one conv.  dc comes out of a BatchNorm
    backward (_bn_bwd*): either an ops.BnApply -- the recipe of dc, evaluated by the GEMM loaders -- or a tensor carrying
    max |dc| (dc._koaf_amax); both put the two contractions on the fp16 scheme (koaf.h: KoafGemm.fmt 1)."""
    w = packed_weight(conv.weight)
    cin, cout = conv.in_channels, conv.out_channels
    k, s, p, g = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.groups
    wimg = weight_planes(conv.weight) if g == 1 else None
    if isinstance(dc, ops.BnApply) and (g != 1 or (need_dx and wimg is None)):
        dc = dc.materialize(want_amax=True)            # grouped 3x3 / no weight plane images: the element-wise pass (max |dc| rides along:
        #                                                the fp16 scheme of both kinds of consumer)
    amax = getattr(dc, "_koaf_amax", None)
    sc, sh = (in_saved[2], in_saved[3]) if in_saved is not None else (None, None)
    gw, acc = grad_target(conv.weight)

    def wgrad():
        if g == 1:
            ops.conv2d_wgrad(dc, x, gw, N, H, W, cin, cout, k, k, s, p, sc, sh, dy_amax=amax)
        else:
            dwexp = ops.gconv3x3_wgrad(dc, x, N, H, W, cin, s, sc, sh)
            ops.gconv_compress_dw(dwexp, gw, cin, g)
    if side is None:
        wgrad()
        deliver_grad(conv.weight, gw, acc)
    early = side is not None and WGRAD_EARLY and need_dx and g == 1 and k == 1
    if early:
        # (A/B switch KOAF_WGRAD_EARLY=1: the 1x1 weight gradient starts TOGETHER with its sibling data gradient -- both stream
        # through the same (dz, c) tensors, and a second reader close behind the first is served by the 256 MB Infinity Cache)
        held = dc.tensors() if isinstance(dc, ops.BnApply) else (dc, amax)
        side.run(held + (x, in_saved), wgrad, conv.weight, gw, acc)
    dx = None
    if need_dx:
        if g == 1:
            dx = ops.conv2d_dgrad(dc, w, N, H, W, cin, cout, k, k, s, p, residual=residual, bnb=bnb, wimg=wimg, dy_amax=amax)
        else:
            assert residual is None and bnb is None
            dx = ops.gconv3x3_dgrad(dc, wexp, N, H, W, cin, s)
    if side is not None and not early:
        # enqueued BEHIND the sibling dgrad: the side stream starts this wgrad when the dgrad is done, so it
        # overlaps the HBM-bound BatchNorm backward of the next layer instead of fighting the dgrad for MFMAs
        held = dc.tensors() if isinstance(dc, ops.BnApply) else (dc, amax)
        side.run(held + (x, in_saved), wgrad, conv.weight, gw, acc)
    return dx


def _bn_bwd(bn, g, c, saved, rows, mask_mode, ymask=None, dz_out=None, dc_out=None, fused=None, pool=None):
    """-> dc of a BatchNorm(+ReLU mask): an ops.BnApply when the consumers can form it on load (fused), else the tensor"""
    C = bn.num_features
    gg, ag = grad_target(bn.weight)
    gb, ab = grad_target(bn.bias)
    if fused is None:
        fused = ops.CONV_F16
        dc = ops.bn_bwd(g, c, saved, rows, C, rows, gg, gb, mask_mode, ymask=ymask, dz_out=dz_out, fused=fused)
        if fused and not FUSE_APPLY:
            dc = dc.materialize(out=dc_out, want_amax=True)      # (A/B switch: dc written out, with its exact max |dc|)
    else:
        dc = ops.bn_bwd(g, c, saved, rows, C, rows, gg, gb, mask_mode, ymask=ymask, dz_out=dz_out,
                        dc_out=None if fused else dc_out, fused=fused, pool=pool)
    deliver_grad(bn.weight, gg, ag)
    deliver_grad(bn.bias, gb, ab)
    return dc


def _bn_bwd_part(bn, part, nsum, i1, dz, c, saved, rows, dc_out=None, dzmax=None):
    """BatchNorm backward whose reduction was fused into the producing dgrad's epilogue (dzmax: the max |dz| it left)."""
    C = bn.num_features
    gg, ag = grad_target(bn.weight)
    gb, ab = grad_target(bn.bias)
    fused = ops.CONV_F16 and dzmax is not None
    dc = ops.bn_bwd_from_part(part, nsum, i1, dz, c, saved, rows, C, rows, gg, gb, dc_out=None if fused else dc_out,
                              fused=fused, dzmax=dzmax)
    if fused and not FUSE_APPLY:
        dc = dc.materialize(out=dc_out, want_amax=True)
    deliver_grad(bn.weight, gg, ag)
    deliver_grad(bn.bias, gb, ab)
    return dc


FUSE_BNB = os.environ.get("KOAF_FUSE_BNB", "1") != "0"
FUSE_STEM_BWD = os.environ.get("KOAF_FUSE_STEM_BWD", "1") != "0"
REUSE_STAGE_INPUT = os.environ.get("KOAF_REUSE_STAGE_INPUT", "1") != "0"


def _tail_bnb(prev):
    """epilogue descriptor for the tail (relu(bn(c_last) + identity)) of block record `prev`"""
    if prev is None or not FUSE_BNB:
        return None
    c_last, s_last = (prev.c3, prev.s3) if prev.kind == "bottleneck" else (prev.c2, prev.s2)
    d = dict(mode=1, c=c_last, y=prev.y, saved=s_last, dz_amax=ops.CONV_F16)
    if prev.cd is not None:
        d["c2"], d["saved2"] = prev.cd, prev.sd
    return d


def _block_fwd(blk, y, N, Hc, Wc, train, given, tail=None, defer=False, skip_tail=False):
    """Forward of one residual block.  given = None: statistics are collected by the conv epilogues and
    finalised (normal forward).  given = (s1, s2, s3, sd): activation RECOMPUTE in backward -- the saved
    BatchNorm statistics are reused, nothing is reduced and no running statistic is touched.
    tail = (c_last, s_last, identity, y buffer, identity's BatchNorm record or None) of the PREVIOUS block whose tail was deferred (y is None then): this block's conv1 forms
    its own input on load and writes it (r.yin).  defer: leave THIS block's tail to the next block (r.y stays None, r.tail
    holds what the next call needs); the caller checks _can_take_tail(next block).  skip_tail: the caller already holds this
    block's output (r.y stays None, nothing is computed for it)."""
    r = _Rec()
    r.blk, r.yin = blk, y
    want = train and given is None

    def fin(bn, part, count, idx):
        return given[idx] if given is not None else _bn_fin(bn, part, count)
    if isinstance(blk, Bottleneck):
        r.kind = "bottleneck"
        # bn1 known before conv1 runs (rebuild: saved statistics; eval: running statistics): conv1's epilogue cuts conv2's
        # plane images (same kernels, same bits -- test_stage_recompute_matches_stored_activations)
        s1_pre, emit = None, None
        if not want and _takes_planes(blk.conv2) and blk.conv1.groups == 1 and weight_planes(blk.conv1.weight) is not None:
            s1_pre = fin(blk.bn1, None, N * Hc * Wc, 0)
            emit = (s1_pre[2], s1_pre[3])
        if tail is not None:
            r.c1, part, _, _, _, y = _conv_fwd(tail[0], blk.conv1, N, Hc, Wc, tail[1], want, blk.bn1, tail_idt=tail[2], tail_out=tail[3],
                                               tail_idsaved=tail[4], emit=emit)
            r.yin = y
        else:
            r.c1, part, _, _, _ = _conv_fwd(y, blk.conv1, N, Hc, Wc, None, want, blk.bn1, emit=emit)
        r.s1 = s1_pre if s1_pre is not None else fin(blk.bn1, part, N * Hc * Wc, 0)
        # (rebuilt in backward: the plane images cut for conv2 live on until its weight gradient, a few kernels later, instead of
        # being cut again -- in the forward pass proper they would have to survive the whole step)
        r.c2, part, OH, OW, r.wexp = _conv_fwd(r.c1, blk.conv2, N, Hc, Wc, r.s1, want, blk.bn2,
                                               keep_planes=given is not None and KEEP_PLANES_RECOMPUTE)
        r.s2 = fin(blk.bn2, part, N * OH * OW, 1)
        r.c3, part, _, _, _ = _conv_fwd(r.c2, blk.conv3, N, OH, OW, r.s2, want, blk.bn3)
        r.s3 = fin(blk.bn3, part, N * OH * OW, 2)
        last_c, last_s, cout = r.c3, r.s3, blk.conv3.out_channels
    elif isinstance(blk, BasicBlock):
        r.kind = "basic"
        r.c1, part, OH, OW, _ = _conv_fwd(y, blk.conv1, N, Hc, Wc, None, want, blk.bn1)
        r.s1 = fin(blk.bn1, part, N * OH * OW, 0)
        r.c2, part, _, _, _ = _conv_fwd(r.c1, blk.conv2, N, OH, OW, r.s1, want, blk.bn2)
        r.s2 = fin(blk.bn2, part, N * OH * OW, 1)
        last_c, last_s, cout = r.c2, r.s2, blk.conv2.out_channels
    else:
        raise TypeError(f"unsupported block {type(blk)}")
    rows_o = N * OH * OW
    if blk.downsample is not None:
        r.cd, part, _, _, _ = _conv_fwd(y, blk.downsample[0], N, Hc, Wc, None, want, blk.downsample[1])
        r.sd = fin(blk.downsample[1], part, rows_o, 3)
        if skip_tail:
            pass
        elif defer and FUSE_TAIL_DS:
            r.tail = (last_c, last_s, r.cd, torch.empty_like(last_c), r.sd)     # (identity = bn_d(cd), formed on load as well)
        else:
            r.y = ops.bn_add_relu(last_c, last_s, rows_o, cout, idt=r.cd, idsaved=r.sd)
    elif skip_tail:
        pass
    elif defer:
        # y = relu(bn(last_c) + identity) is formed (and written) by the next block's conv1.  Its buffer is allocated HERE, where
        # the element-wise pass would have allocated its output: the caching allocator then sees the same request order as
        # without the fusion (allocating it at the next conv1 instead cost 17 GB of reserved memory on the headline step)
        r.tail = (last_c, last_s, y, torch.empty_like(last_c), None)
    else:
        r.y = ops.bn_add_relu(last_c, last_s, rows_o, cout, idt=y)
    r.dims = (N, Hc, Wc, OH, OW)
    return r


def _blocks_fwd(blocks, y, N, Hc, Wc, train, givens=None, tail=None, defer_last=False, slim=False, last_y=None):
    """forward of consecutive blocks with the bottleneck tails left to the following conv1 where possible; -> (records, y of the
    last block or None when its tail was deferred (records[-1].tail), H, W).  tail: a deferred tail entering the first block.
    slim: the records are not needed for backward (a stage that will be rebuilt, or no gradient at all): each one gives up its
    conv outputs as soon as the following block has consumed them, so that only ~two blocks are alive at a time.
    last_y: the output of the last block, when the caller still holds it (a stage rebuilt in backward: it is the saved input of
    the following stage) -- its tail is then not computed again."""
    recs, yin0 = [], None
    for i, blk in enumerate(blocks):
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        defer = (nxt is not None and _can_take_tail(nxt)) or (nxt is None and defer_last)
        have = nxt is None and last_y is not None
        r = _block_fwd(blk, y, N, Hc, Wc, train, givens[i] if givens is not None else None, tail=tail, defer=defer, skip_tail=have)
        if have:
            r.y = last_y
        if i == 0:
            yin0 = r.yin                     # the input of the first block (written by its conv1 when a tail came in)
        if recs:
            if tail is not None:
                recs[-1].y = r.yin           # the previous block's output, written by this block's conv1
            if slim:
                p = recs[-1]
                p.c1 = p.c2 = p.c3 = p.cd = p.y = p.yin = p.tail = p.wexp = None
        recs.append(r)
        y, Hc, Wc = r.y, r.dims[3], r.dims[4]
        tail = r.tail
    return recs, y, Hc, Wc, yin0


def _recompute_plan(trunk, nstages):
    """KoafTrunk.recompute -> (stages to rebuild in backward, one block at a time?).  False: every conv output is kept;
    True: every stage is rebuilt; "block": every stage, one block at a time (least memory, one more forward of most
    blocks); a collection of stage indices (0 = layer1): only those stages -- the early stages hold most of the bytes
    (80 of the 180 MB per 384^2 slice sit in layer1, 10 in layer4) for about the same FLOPs as the late ones, so
    rebuilding layer1-2 and keeping layer3-4 buys most of the memory for half of the recompute time."""
    r = getattr(trunk, "recompute", False)
    if r is False or r is None:
        return frozenset(), False
    if r is True or r == "stage":
        return frozenset(range(nstages)), False
    if r == "block":
        return frozenset(range(nstages)), True
    rs = frozenset(int(i) for i in r)
    if not rs <= frozenset(range(nstages)):
        raise ValueError(f"KoafTrunk.recompute: stage indices out of range 0..{nstages - 1}: {sorted(rs)}")
    return rs, False


class EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trunk, anchor, keep, lane):
        st = trunk._koaf_layout()
        if x.dim() == 4:
            if x.shape[1] != 1:
                raise ValueError("KoafTrunk takes the single-channel image (the 1->3 repeat is folded into conv1)")
            N, _, H, W = x.shape
        else:
            N, H, W = x.shape
        x = x.contiguous()
        conv1, bn1 = st["conv1"], st["bn1"]
        train = bn1.training
        w1t = ops.stem_fold_w(packed_weight(conv1.weight))
        adt = getattr(trunk, "act_dtype", torch.float32)       # storage type of the forward activations (KoafTrunk.act_dtype)
        if train:       # (the statistics of bn1 come out of the stem kernel: no pass over its output, the largest tensor of the trunk)
            c0, part = ops.stem_fwd(x, w1t, N, H, W, dtype=adt, stats=True, shift=_stat_shift(bn1, train))
        else:
            c0, part = ops.stem_fwd(x, w1t, N, H, W, dtype=adt), None
        H1, W1 = c0.shape[1], c0.shape[2]
        s0 = _bn_fin(bn1, part, N * H1 * W1)
        y, am = ops.maxpool_fwd(c0, s0, N, H1, W1, 64)
        Hc, Wc = y.shape[1], y.shape[2]
        # per stage: either the block records (every conv output) are kept for backward, or -- activation recompute --
        # only the stage input and the BatchNorm statistics, and the stage is rebuilt right before its backward
        rset, block_level = _recompute_plan(trunk, len(st["stages"]))
        if not keep:
            rset = frozenset()
        stages_saved = []
        tail, prev_recs = None, None
        nst = len(st["stages"])
        for si, stage in enumerate(st["stages"]):
            rec_stage = si in rset
            # (the last block's tail of this stage is left to the first conv1 of the next stage where that one can take it)
            defer_last = si + 1 < nst and _can_take_tail(st["stages"][si + 1][0])
            recs, y_out, Ho, Wo, yin0 = _blocks_fwd(stage, y, N, Hc, Wc, train, tail=tail, defer_last=defer_last, slim=rec_stage or not keep)
            if tail is not None:
                y = yin0                              # this stage's input: written by its first conv1
                if prev_recs:
                    prev_recs[-1].y = y               # = the previous stage's last block output
            sv = dict(yin=y if rec_stage else None, H=Hc, W=Wc, stats=[], recs=[], recompute=rec_stage)
            if rec_stage:
                sv["stats"] = [(r.s1, r.s2, r.s3, r.sd) for r in recs]
            elif keep:
                sv["recs"] = recs
            tail = recs[-1].tail
            prev_recs = sv["recs"]
            if rec_stage or not keep:
                for r in recs:                        # (drop what is left of the conv outputs; a deferred tail holds what it needs)
                    r.c1 = r.c2 = r.c3 = r.cd = r.yin = r.y = None
            del recs
            y, Hc, Wc = y_out, Ho, Wo
            stages_saved.append(sv)
        C = y.shape[-1]
        if st["gap"]:
            out = ops.gap_fwd(y, N, Hc * Wc, C).view(N, C, 1, 1)
        else:
            out = y.permute(0, 3, 1, 2)  # (N,C,h,w) view of the NHWC buffer
            if out.dtype != torch.float32:
                out = out.float()        # (what leaves the trunk is fp32; only the trunk's own activations are stored as bf16)
        if keep:
            if 0 in rset:
                c0 = None          # the stem output is rebuilt in backward too (one cheap 7x7 conv; 64 x H/2 x W/2 floats)
            ctx.state = dict(x=x, c0=c0, s0=s0, am=am, stages=stages_saved, any_recompute=bool(rset), caller=CALLER_STREAM,
                             block_level=block_level, adt=adt,
                             dims=(N, H, W, H1, W1), last=(Hc, Wc, C), st=st, lane=lane, train=train)
        return out

    @staticmethod
    def backward(ctx, gout):
        S = ctx.state
        ctx.state = None
        lane = S["lane"]
        if lane is None:
            return EncoderFn._backward_body(S, gout, None)
        # multi-stream: this encoder's backward runs on its own lane (autograd already switched to the forward
        # stream and ordered it after the producer of gout); the caller's stream joins at the end of backward()
        main_s, side_s = lane_streams(gout.device, lane)
        cur = torch.cuda.current_stream()
        if cur != main_s:
            main_s.wait_stream(cur)
            gout.record_stream(main_s)
        with torch.cuda.stream(main_s):
            out = EncoderFn._backward_body(S, gout, side_s)
        _register_join(main_s, S.get("caller"))
        return out

    @staticmethod
    def _blocks_bwd(recs, dy, side):
        """backward through a list of block records (last first); returns the gradient w.r.t. the first block's input"""
        pend = None   # (part, nsum) of THIS block's tail when the previous dgrad's epilogue already reduced it
        while recs:
            r = recs.pop()
            prev = recs[-1] if recs else None
            blk = r.blk
            N, Hi, Wi, OH, OW = r.dims
            rows_o, rows_i = N * OH * OW, N * Hi * Wi
            bott = r.kind == "bottleneck"
            tail_bn = blk.bn3 if bott else blk.bn2
            tail_c, tail_s = (r.c3, r.s3) if bott else (r.c2, r.s2)
            # tail: dz = dy*[y>0] and the BatchNorm behind it
            if pend is None:
                dcl = _bn_bwd(tail_bn, dy, tail_c, tail_s, rows_o, 1, ymask=r.y, dz_out=dy)
            else:
                dcl = _bn_bwd_part(tail_bn, pend[0], pend[1], 1, dy, tail_c, tail_s, rows_o, dzmax=pend[2])
            dz = dy
            inner = FUSE_BNB
            want_max = ops.CONV_F16

            def split(res):
                """(dz, part, dzmax or None) of a dgrad that carried a fused BatchNorm-backward reduction"""
                return res if len(res) == 3 else (res[0], res[1], None)
            if bott:
                g2 = blk.conv2.groups == 1 and inner
                res = _conv_bwd(blk.conv3, dcl, r.c2, N, OH, OW, r.s2, None, side=side,
                                bnb=dict(mode=2, c=r.c2, saved=r.s2, dz_amax=want_max) if inner else None)
                del dcl
                if inner:
                    da2, part2, mx2 = split(res)
                    dc2 = _bn_bwd_part(blk.bn2, part2, 2, 1, da2, r.c2, r.s2, rows_o, dc_out=da2, dzmax=mx2)
                else:
                    da2 = res
                    dc2 = _bn_bwd(blk.bn2, da2, r.c2, r.s2, rows_o, 2, dc_out=da2)
                res = _conv_bwd(blk.conv2, dc2, r.c1, N, Hi, Wi, r.s1, r.wexp, side=side,
                                bnb=dict(mode=2, c=r.c1, saved=r.s1, dz_amax=want_max) if g2 else None)
                del dc2, da2
                if g2:
                    da1, part1, mx1 = split(res)
                    dc1 = _bn_bwd_part(blk.bn1, part1, 2, 1, da1, r.c1, r.s1, rows_i, dc_out=da1, dzmax=mx1)
                else:
                    da1 = res
                    dc1 = _bn_bwd(blk.bn1, da1, r.c1, r.s1, rows_i, 2, dc_out=da1)
            else:
                res = _conv_bwd(blk.conv2, dcl, r.c1, N, OH, OW, r.s1, None, side=side,
                                bnb=dict(mode=2, c=r.c1, saved=r.s1, dz_amax=want_max) if inner else None)
                del dcl
                if inner:
                    da1, part1, mx1 = split(res)
                    dc1 = _bn_bwd_part(blk.bn1, part1, 2, 1, da1, r.c1, r.s1, rows_o, dc_out=da1, dzmax=mx1)
                else:
                    da1 = res
                    dc1 = _bn_bwd(blk.bn1, da1, r.c1, r.s1, rows_o, 2, dc_out=da1)
            first = blk.conv1
            # identity branch
            if blk.downsample is not None:
                if pend is None:
                    dcd = _bn_bwd(blk.downsample[1], dz, r.cd, r.sd, rows_o, 0, dc_out=dz)
                else:
                    dcd = _bn_bwd_part(blk.downsample[1], pend[0], pend[1], 2, dz, r.cd, r.sd, rows_o, dc_out=dz, dzmax=pend[2])
                resid = _conv_bwd(blk.downsample[0], dcd, r.yin, N, Hi, Wi, None, None, side=side)
            else:
                resid = dz
            # block-entry dgrad (+identity gradient); its epilogue reduces the PREVIOUS block's tail BatchNorm(s)
            bnb = _tail_bnb(prev)
            res = _conv_bwd(first, dc1, r.yin, N, Hi, Wi, None, None, residual=resid, side=side, bnb=bnb)
            if bnb is not None:
                dy, part, mx = split(res)
                pend = (part, 3 if "c2" in bnb else 2, mx)
            else:
                dy, pend = res, None
            del dc1, da1, dz, resid, r
            if side is not None and side.hold:
                side.join()        # (recompute: the block's tensors may die now, see _SideStream)
        return dy

    @staticmethod
    def _backward_body(S, gout, side_stream):
        st = S["st"]
        N, H, W, H1, W1 = S["dims"]
        Hc, Wc, C = S["last"]
        if st["gap"]:
            dy = ops.gap_bwd(gout.reshape(N, C).contiguous(), N, Hc * Wc, C).view(N, Hc, Wc, C)
        else:
            dy = gout.permute(0, 2, 3, 1).contiguous()
            if dy.data_ptr() == gout.data_ptr():
                dy = dy.clone()  # masked in place below
        side = _SideStream(gout.device, side_stream, hold=S["any_recompute"]) if USE_SIDE_STREAM else None
        stages = st["stages"]
        run = []      # block records of consecutive kept stages: one list, so the fused tail reductions cross stage boundaries
        next_in = None   # the input of the stage processed last = the output of the stage processed next (when still held)
        for si in range(len(stages) - 1, -1, -1):
            sv = S["stages"][si]
            if not sv["recompute"]:
                run = sv["recs"] + run
                sv["recs"] = None
                if si > 0 and not S["stages"][si - 1]["recompute"]:
                    continue
                first_in = run[0].yin if REUSE_STAGE_INPUT else None
                dy = EncoderFn._blocks_bwd(run, dy, side)
                run = []
                next_in = first_in
                continue
            # activation recompute: only the stage input and the BatchNorm statistics were kept; the stage's conv
            # outputs are rebuilt (same kernels, saved statistics, no reductions) right before use
            y, Hc2, Wc2 = sv["yin"], sv["H"], sv["W"]
            sv["yin"] = None
            stage_in, last_y = (y if REUSE_STAGE_INPUT else None), next_in
            next_in = None
            if S["block_level"] and len(stages[si]) > 1:
                # block-granular: pass 1 rebuilds only the block INPUTS of the stage, then every block is rebuilt
                # alone (last first) and back-propagated -- one block's conv outputs live at a time instead of
                # the whole stage's, for one more forward of the stage's blocks but the last
                ins = [(y, Hc2, Wc2)]
                for blk, given in zip(stages[si][:-1], sv["stats"][:-1]):
                    r = _block_fwd(blk, y, N, Hc2, Wc2, S["train"], given)
                    y, Hc2, Wc2 = r.y, r.dims[3], r.dims[4]
                    ins.append((y, Hc2, Wc2))
                    del r
                for bi in range(len(stages[si]) - 1, -1, -1):
                    yb, hb, wb = ins.pop()
                    r = _block_fwd(stages[si][bi], yb, N, hb, wb, S["train"], sv["stats"][bi])
                    dy = EncoderFn._blocks_bwd([r], dy, side)
                    del r, yb
                del y, last_y
                next_in = stage_in
                continue
            # (the stage's output is the input the following stage kept for its own backward: the last tail is not rebuilt)
            recs, y, Hc2, Wc2, _ = _blocks_fwd(stages[si], y, N, Hc2, Wc2, S["train"], givens=sv["stats"], last_y=last_y)
            del last_y
            dy = EncoderFn._blocks_bwd(recs, dy, side)
            del recs, y
            next_in = stage_in
        # stem: max-pool, BN0, conv1 weight gradient (no data gradient: the input is a leaf)
        next_in = None
        conv1, bn1 = st["conv1"], st["bn1"]
        c0 = S["c0"]
        if c0 is None:
            c0 = ops.stem_fwd(S["x"], ops.stem_fold_w(packed_weight(conv1.weight)), N, H, W, dtype=S["adt"])
        if FUSE_STEM_BWD:
            # the max-pool's input gradient is gathered inside the BatchNorm reduction and dc0 is formed by the weight gradient
            # while it loads (dz, c0): neither tensor is written (two of the four passes over the largest activation of the trunk)
            dc0 = _bn_bwd(bn1, None, c0, S["s0"], N * H1 * W1, 2, fused=True, pool=(dy, S["am"], N, H1, W1))
        else:
            da0 = ops.maxpool_bwd(dy, S["am"], N, H1, W1, 64)
            dc0 = _bn_bwd(bn1, da0, c0, S["s0"], N * H1 * W1, 2, dc_out=da0, fused=False)
        gw, acc = grad_target(conv1.weight)
        ops.stem_wgrad(dc0, S["x"], gw, N, H, W)
        deliver_grad(conv1.weight, gw, acc)
        if side is not None:
            side.join()
        return None, None, None, None, None


class KoafTrunk(nn.Sequential):
    """`nn.Sequential(*children)` with the reference's child indices (state_dict keys `0.weight`,
    `1.running_mean`, `4.0.conv1.weight`, ...), executed as the fused HIP schedule above.

    forward(x): x = the SINGLE-channel image batch (N,1,H,W); returns (N,C,1,1) with the GAP child
    present, else (N,C,h,w).

    act_dtype: how the trunk's forward activations (stem / conv / block outputs) are STORED in HBM: torch.float32 (default,
    the parity mode) or torch.bfloat16 (koaf.h "bf16 ACTIVATION STORAGE": half the bytes of every HBM-bound kernel and of the
    tensors saved for backward; arithmetic, statistics, gradients and parameters stay fp32).  Set per model by the config key
    `activation_storage: bf16` (models/_common.apply_activation_storage)."""
    act_dtype = torch.float32

    def _koaf_layout(self):
        lay = self.__dict__.get("_koaf_lay")
        if lay is not None:
            return lay
        ch = list(self.children())
        if not (len(ch) >= 8 and isinstance(ch[0], nn.Conv2d) and isinstance(ch[1], nn.BatchNorm2d)
                and isinstance(ch[3], nn.MaxPool2d)):
            raise TypeError("KoafTrunk expects the children of a koaf ResNet (conv1, bn1, relu, maxpool, layer1-4[, avgpool])")
        c1 = ch[0]
        if c1.kernel_size != (7, 7) or c1.stride != (2, 2) or c1.padding != (3, 3) or c1.out_channels != 64:
            raise NotImplementedError("stem other than 7x7/s2/p3 -> 64 is not built")
        blocks, stages = [], []
        gap = False
        for m in ch[4:]:
            if isinstance(m, nn.Sequential):
                blocks.extend(list(m.children()))
                stages.append(list(m.children()))
            elif isinstance(m, nn.AdaptiveAvgPool2d):
                gap = True
            else:
                raise TypeError(f"unexpected trunk child {type(m)}")
        lay = dict(conv1=ch[0], bn1=ch[1], blocks=blocks, stages=stages, gap=gap)
        self.__dict__["_koaf_lay"] = lay
        return lay

    def forward(self, x, lane=None):
        lay = self._koaf_layout()
        anchor = lay["conv1"].weight
        # autograd runs Function.forward with grad mode off, so decide here whether backward can happen
        keep = torch.is_grad_enabled() and anchor.requires_grad
        return EncoderFn.apply(x, self, anchor, keep, lane)
