"""On-device "last-chance" downscale (reference: koafusion/preproc/_pt.py:175-200, called from
koafusion/run/train_prog_fus.py:111-116).  F.interpolate(scale 0.5, recompute_scale_factor=True,
align_corners=False, linear/bilinear/trilinear) on even sizes is exactly 2x average pooling -- one
HBM-bound HIP kernel here."""
from .. import ops


class PTInterpolate(object):
    def __init__(self, scale_factor):
        self.scale_factor = tuple(scale_factor) if not isinstance(scale_factor, (int, float)) else (scale_factor,)

    def __call__(self, image, mask=None):
        """image: (B, CH, D0[, D1[, D2]]) -- returns the resized image (mask is not built)"""
        if mask is not None:
            raise NotImplementedError("mask resizing is not on the train path")
        sf = self.scale_factor
        if all(float(s) == 1.0 for s in sf):
            return image
        if image.shape[1] != 1:
            raise NotImplementedError("single-channel inputs only")
        x = image.contiguous()
        if image.ndim == 4 and tuple(map(float, sf)) == (0.5, 0.5):
            B, _, R, C = image.shape
            return ops.downscale2(x, B, R, C, 1, 1).view(B, 1, R // 2, C // 2)
        if image.ndim == 5 and tuple(map(float, sf[:2])) == (0.5, 0.5) and float(sf[2]) in (0.5, 1.0):
            B, _, R, C, S = image.shape
            fs = 2 if float(sf[2]) == 0.5 else 1
            return ops.downscale2(x, B, R, C, S, fs).view(B, 1, R // 2, C // 2, S // fs)
        raise NotImplementedError(f"scale_factor {sf} is not built (the recipes use 0.5 / 1.0 only, runner.sh:347-361)")
