"""Pieces shared by the six registry models."""
import torch
from torch import nn

from .. import functional as KF
from .. import ops
from ..arena import get_arena
from ._core_fes import dict_fes
from ._encoder import KoafTrunk, lane_streams


def build_trunk(arch, pretrained, with_gap):
    """`nn.Sequential(*list(fe.children())[:-1 or :-2])` of the reference (e.g. _xrNmrMcP.py:40-59)."""
    fe = dict_fes[arch](pretrained=pretrained)
    ch = list(fe.children())
    ch = ch[:-1] if with_gap else ch[:-2]
    return KoafTrunk(*ch)


class KoafDropout2d(nn.Module):
    """nn.Dropout2d on the encoder output (_xrNmrMcP.py:62-72): one draw per (image, channel) on the libkoaf
    counter-based generator -- element-wise on the pooled (N,C,1,1) output, koaf_dropout2d on a spatial one."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        if x.shape[2] != 1 or x.shape[3] != 1:
            return KF.dropout2d(x, self.p, True)        # spatial (with_gap=false) output: whole channels dropped
        return KF.dropout(x, self.p, True)

    def extra_repr(self):
        return f"p={self.p}"


def make_drop(p):
    return KoafDropout2d(p) if p else nn.Identity()


def mr_view(config):
    """slice view of the MRI volumes of a fusion model: "rc" on the reference's (B,1,R,C,S) tensors, or -- config key
    `fe.mr.volume_layout: ncdhw` (not in the reference; default `rcs`) -- "src": the volumes arrive slice-major
    (B,1,S,R,C), as BASELINE.json writes them (1x160x384x384), and the slice fold is a zero-copy view."""
    fe = config["fe"]
    lay = dict(fe["mr"] if "mr" in fe else fe).get("volume_layout", "rcs")     # (MR-only models keep the keys under `fe`)
    if lay not in ("rcs", "ncdhw"):
        raise ValueError("Unsupported `model.fe.mr.volume_layout` (rcs | ncdhw)")
    return "src" if lay == "ncdhw" else "rc"


def fold_slices(x, dims_view="rc"):
    """(B,1,R,C,S) -> single-channel image batch for the 2-D trunk (the 1->3 repeat is folded into conv1).
    rc: "b ch r c s -> (b s) ch r c" (_xrNmrMcP.py:209); cs / rs: _mrN_cnn_trf.py:112-117; src: the same slices as rc
    from a slice-major (B,1,S,R,C) volume (see mr_view) -- no data movement."""
    if dims_view == "src":
        B, ch, S, R, C = x.shape
        if ch != 1:
            raise ValueError("koafusion volumes are single-channel")
        return x.contiguous().view(B * S, 1, R, C)
    B, ch, R, C, S = x.shape
    if ch != 1:
        raise ValueError("koafusion volumes are single-channel")
    x = x.contiguous()
    if dims_view == "rc":
        return ops.slice_fold(x, B, R, C, S).view(B * S, 1, R, C)
    if dims_view == "cs":
        return x.view(B * R, 1, C, S)
    if dims_view == "rs":
        return x.view(B, R, C, S).permute(0, 2, 1, 3).contiguous().view(B * C, 1, R, S)
    raise ValueError("Unsupported `model.fe.dims_view`")


def tokens(feat, B):
    """"(b s) ch d0 d1 -> b (s d0 d1) ch" on the trunk output (an NHWC buffer viewed as NCHW)."""
    N, C, h, w = feat.shape
    t = feat.permute(0, 2, 3, 1)            # (N,h,w,C): the memory order
    return t.reshape(B, (N // B) * h * w, C)


def adopt(model, *inputs):
    for t in inputs:
        if not t.is_cuda:
            raise RuntimeError("koaf models run on a HIP device only (no CPU fallback); move model and inputs to cuda")
    return get_arena(model)


def finish(config, res_out):
    from collections import OrderedDict
    endpoints = OrderedDict()
    endpoints["main"] = res_out
    if config.output_type == "main":
        return endpoints["main"]
    elif config.output_type == "dict":
        return endpoints
    raise ValueError(f"Unknown output_type: {config.output_type}")


def apply_activation_storage(model, config):
    """config key `activation_storage` (not in the reference; default "fp32"): "bf16" stores the forward activations of every
    encoder trunk as bf16 (KoafTrunk.act_dtype) -- BASELINE.json config 2's "bf16", a throughput mode reported beside the fp32
    parity mode with its measured error"""
    try:
        mode = config["activation_storage"]
    except (KeyError, AttributeError):
        mode = "fp32"
    if mode not in ("fp32", "bf16"):
        raise ValueError("Unsupported `model.activation_storage` (fp32 | bf16)")
    for m in model.modules():
        if isinstance(m, KoafTrunk):
            m.act_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
    return mode


def maybe_restore(model, config, path_weights):
    apply_activation_storage(model, config)
    if config["restore_weights"]:
        model.load_state_dict(torch.load(path_weights, map_location="cpu"))


MAPPING_CH = {"resnet18": 512, "resnet34": 512, "resnet50": 2048, "resnext50_32x4d": 2048}
MAPPING_SPAT = {320: 10, 160: 5, 128: 4, 96: 3, 64: 2, 32: 1, 350: 11, 25: 1}


import os

USE_LANES = os.environ.get("KOAF_ENCODER_LANES", "1") != "0"


def run_trunks(jobs):
    """Run independent encoders concurrently, one HIP stream pair ("lane") each.
    jobs: [(trunk, input, dims_view or None[, post])] -- dims_view None = 2-D radiograph, else the MRI slice fold;
    `post(feature_map)` (optional) runs on the same lane right after the encoder (the per-MRI aggregator of the
    hierarchical models: small-grid kernels that overlap the other lanes instead of running alone after the join).
    The encoders share nothing but read-only inputs, so their ~160 kernels each interleave on the 256 CUs: one
    lane's HBM-bound BatchNorm kernels and GEMM tails are filled by the other lanes' MFMA blocks.  The caller's
    stream waits for all lanes before the results are consumed; in backward each node replays on its lane."""
    def one(job, lane):
        trunk, x, view = job[0], job[1], job[2]
        post = job[3] if len(job) > 3 else None
        xin = fold_slices(x, view) if view is not None else x
        out = trunk(xin, lane=lane)
        return post(out) if post is not None else out
    # activation recompute trades memory for FLOPs; concurrent lanes would hold several stages' rebuilt
    # activations at once and defeat it, so recomputing trunks run one after the other
    if not USE_LANES or len(jobs) < 2 or any(getattr(j[0], "recompute", False) for j in jobs):
        return [one(j, None) for j in jobs]
    from . import _encoder
    main = torch.cuda.current_stream()
    ev = main.record_event()
    outs = []
    dev = jobs[0][1].device
    _encoder.CALLER_STREAM = main          # (the lanes' backward joins this stream at the end of backward())
    try:
        for lane, job in enumerate(jobs):
            s, _ = lane_streams(dev, lane)
            s.wait_event(ev)
            with torch.cuda.stream(s):
                out = one(job, lane)
            job[1].record_stream(s)
            out.record_stream(main)
            outs.append(out)
    finally:
        _encoder.CALLER_STREAM = None
    for lane in range(len(jobs)):
        main.wait_stream(lane_streams(dev, lane)[0])
    return outs
