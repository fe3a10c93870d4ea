"""On-device "last-chance" downscale (reference: koafusion/preproc/_pt.py:175-200, called from
koafusion/run/train_prog_fus.py:111-116).  F.interpolate(scale 0.5, recompute_scale_factor=True,
align_corners=False, linear/bilinear/trilinear) on even sizes is exactly 2x average pooling -- one
HBM-bound HIP kernel here."""
import math
import random

import torch

from .. import ops


class PTInterpolate(object):
    """F.interpolate(image, scale_factor, recompute_scale_factor=True, align_corners=False, mode = linear | bilinear |
    trilinear by rank) exactly as the reference's transform (preproc/_pt.py:175-200), for any scale factor: output size
    floor(in * scale) per dimension, coordinates from in / out.  The recipes' factors (0.5, 0.5[, 0.5 | 1.0]) on even
    single-channel inputs take the 2x-average-pooling kernel (identical values).  A mask raises ValueError, as in the
    reference: its mask branch hands align_corners=False to mode="nearest", which torch refuses (fixture F8 records it)."""

    def __init__(self, scale_factor):
        self.scale_factor = tuple(scale_factor) if not isinstance(scale_factor, (int, float)) else scale_factor

    def _factors(self, nd):
        sf = self.scale_factor
        if isinstance(sf, (int, float)):
            return (float(sf),) * nd
        if len(sf) != nd:
            raise ValueError(f"scale_factor {sf} does not match the {nd} spatial dimensions of the input")
        return tuple(float(v) for v in sf)

    def __call__(self, image, mask=None):
        """image: (B, CH, D0[, D1[, D2]]) device tensor -> resized image"""
        if image.ndim not in (3, 4, 5):
            raise KeyError(image.ndim)          # (the reference indexes {3: "linear", 4: "bilinear", 5: "trilinear"})
        if mask is not None:
            raise ValueError("align_corners option can only be set with the interpolating modes: linear | bilinear | bicubic | "
                             "trilinear")      # (what the reference's mask branch raises, _pt.py:193-197)
        sf = self._factors(image.ndim - 2)
        if all(s == 1.0 for s in sf):
            return image
        x = image.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        t = image
        if t.shape[1] == 1 and all(s in (0.5, 1.0) for s in sf):
            if t.ndim == 4 and sf == (0.5, 0.5) and t.shape[2] % 2 == 0 and t.shape[3] % 2 == 0:
                B, _, R, C = t.shape
                return ops.downscale2(x, B, R, C, 1, 1).view(B, 1, R // 2, C // 2)
            if t.ndim == 5 and sf[:2] == (0.5, 0.5) and t.shape[2] % 2 == 0 and t.shape[3] % 2 == 0 and (sf[2] == 1.0 or t.shape[4] % 2 == 0):
                B, _, R, C, S = t.shape
                fs = 2 if sf[2] == 0.5 else 1
                return ops.downscale2(x, B, R, C, S, fs).view(B, 1, R // 2, C // 2, S // fs)
        out_size = [int(math.floor(float(n) * s)) for n, s in zip(t.shape[2:], sf)]     # recompute_scale_factor=True
        if min(out_size) < 1:
            raise ValueError(f"scale_factor {sf} empties an input of spatial size {tuple(t.shape[2:])}")
        return ops.resize(x, out_size)


class PTBatchAugment(object):
    """The reference's per-sample tensor pipeline for one image modality
        PTToUnitRange -> PTRotate2D | PTRotate3DInSlice (prob) -> PTGammaCorrection (prob) -> PTNormalize
    (koafusion/datasets/_data_provider.py:295-335, transforms koafusion/preproc/_pt.py:75-345), applied to a whole
    BATCH already resident on the device: one min/max reduction and one fused HBM-bound kernel instead of four CPU
    passes per sample in the loader workers.  Validation / test = the same without the random parts
    (`rotate_prob = gamma_prob = 0`).

    Random state: like the reference's transforms, a rotation draws (p, theta) and a gamma correction (p, gamma) from
    Python's `random` per sample (`randomize()`, _pt.py:228-232, :305-307); `draw(B)` does that for a batch in sample
    order, or pass `states` explicitly ([(p_rot, theta_rad, p_gamma, gamma)] * B)."""

    def __init__(self, mean, std, degree_range=(-15., 15.), rotate_prob=0.5, gamma_range=(0.5, 2.0), gamma_prob=0.5,
                 clip_to_unit=False):
        if clip_to_unit:
            raise NotImplementedError("clip_to_unit=True is not built (the recipes use False)")
        self.mean = float(mean[0] if isinstance(mean, (list, tuple)) else mean)
        self.std = float(std[0] if isinstance(std, (list, tuple)) else std)
        self.theta_range = (math.radians(degree_range[0]), math.radians(degree_range[1]))
        self.rotate_prob, self.gamma_range, self.gamma_prob = rotate_prob, tuple(gamma_range), gamma_prob

    def draw(self, B):
        out = []
        for _ in range(B):
            # a transform that is not in the modality's list draws nothing (T2 maps have no gamma correction,
            # validation has neither: _data_provider.py:304-309,341-360), so the host RNG stream stays the reference's
            p_rot, theta = (random.random(), random.uniform(*self.theta_range)) if self.rotate_prob > 0 else (1.0, 0.0)
            p_gam, gamma = (random.random(), random.uniform(*self.gamma_range)) if self.gamma_prob > 0 else (1.0, 1.0)
            out.append((p_rot, theta, p_gam, gamma))
        return out

    def __call__(self, image, states=None):
        """image: (B, 1, R, C) or (B, 1, R, C, S) raw intensities on the device -> same shape, float32.  Integer volumes
        (uint8 radiographs, uint16 / int16 MRI, as stored on disk) are taken as they are -- 4x / 2x fewer PCIe bytes than
        the fp32 tensors of the reference's loader workers -- and widened on the device (koaf_widen)."""
        if image.ndim not in (4, 5) or image.shape[1] != 1:
            raise ValueError(f"Unsupported tensor shape: {tuple(image.shape)}")
        B, _, R, C = image.shape[:4]
        S = image.shape[4] if image.ndim == 5 else 1
        x = ops.widen(image) if image.dtype in (torch.uint8, torch.uint16, torch.int16) else image.contiguous().float()
        states = self.draw(B) if states is None else list(states)
        if len(states) != B:
            raise ValueError("one (p_rot, theta, p_gamma, gamma) tuple per sample")
        prm = []
        for p_rot, theta, p_gam, gamma in states:
            rot = p_rot < self.rotate_prob
            gam = p_gam < self.gamma_prob
            prm.append([math.cos(theta) if rot else 1.0, math.sin(theta) if rot else 0.0,
                        (1.0 / gamma) if gam else 0.0, 1.0 if rot else 0.0])
        prm = torch.tensor(prm, dtype=torch.float32).to(x.device)
        return ops.augment(x, ops.minmax(x, B), prm, B, R, C, S, self.mean, self.std)


class PinnedPrefetcher(object):
    """Host -> device upload of the next batch while the current step runs (the input side of _data_provider.py:460-498,
    whose DataLoader hands over pageable fp32 tensors): every tensor of a batch dict is staged in a pinned host buffer
    (two sets, reused) and copied on a dedicated HIP stream with non_blocking=True; `next()` makes the caller's stream
    wait for that copy only.  Integer volumes stay integer across PCIe (PTBatchAugment widens them on the device).

        for batch in PinnedPrefetcher(loader, device): ...      # batch: same keys, device tensors (non-tensors pass through)
    """

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PinnedPrefetcher uploads to a HIP device")
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = [{}, {}]
        self._done = [None, None]      # upload-finished event of the batch that last used each pinned set
        self._turn = 0

    def _stage(self, batch):
        turn = self._turn
        bufs, out = self._pinned[turn], {}
        self._turn ^= 1
        if self._done[turn] is not None:
            self._done[turn].synchronize()         # its previous upload has left the pinned buffers
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if not torch.is_tensor(v):
                    out[k] = v
                    continue
                pb = bufs.get(k)
                if pb is None or pb.shape != v.shape or pb.dtype != v.dtype:
                    pb = bufs[k] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                pb.copy_(v)
                out[k] = pb.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self._done[turn] = ev
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._stage(next(it))          # the NEXT batch travels while the caller works on `cur`
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for v in cur.values():
                if torch.is_tensor(v):
                    v.record_stream(torch.cuda.current_stream(self.device))
            yield cur
