"""Loss registry `dict_losses` (reference: koafusion/various/_losses.py:13-117).

FocalLoss / CrossEntropyLoss run forward+backward in one HIP kernel (koaf_focal_loss / koaf_ce_loss).
Kept quirk: FocalLoss ignores batch_avg/class_avg/num_classes and warns about redundant kwargs.
"""
import logging

import torch
from torch import nn

from ..functional import LossFn

logging.basicConfig()
logger = logging.getLogger("losses")
logger.setLevel(logging.DEBUG)


def _weight_on(class_weight, like):
    """class_weight (None | tensor | sequence) as a contiguous fp32 tensor on the logits' device"""
    if class_weight is None:
        return None
    return torch.as_tensor(class_weight, dtype=torch.float32).to(like.device).contiguous()


class CrossEntropyLoss(nn.Module):
    def __init__(self, num_classes, batch_avg=True, batch_weight=None, class_avg=True, class_weight=None, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.batch_avg = batch_avg
        self.class_avg = class_avg
        self.batch_weight = batch_weight
        self.class_weight = class_weight
        logger.warning(f"Redundant loss function arguments:\n{repr(kwargs)}")

    def forward(self, input, target, **kwargs):
        """nn.CrossEntropyLoss(weight=class_weight) on (b, ch[, d0, d1, ...]) logits (_losses.py:36,49)"""
        return LossFn.apply(input, target, 0.0, True, False, _weight_on(self.class_weight, input))


class FocalLoss(nn.Module):
    def __init__(self, num_classes=2, batch_avg=True, batch_weight=None, class_avg=True, class_weight=None, gamma=2,
                 reduction="mean", **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.batch_avg = batch_avg
        self.class_avg = class_avg
        self.batch_weight = batch_weight
        self.class_weight = class_weight
        if reduction not in ("mean", "sum"):
            raise ValueError("Unknown `reduction` value")
        self.reduction = reduction
        self.gamma = gamma
        logger.warning(f"Redundant loss function arguments:\n{repr(kwargs)}")

    def forward(self, input, target, **kwargs):
        """input (b, ch[, d0, d1, ...]) logits, target (b[, d0, d1, ...]) int64 -> scalar: mean|sum over all elements of
        -(1-pt)^gamma * logpt with logpt = -F.cross_entropy(input, target, weight=class_weight, reduction='none')"""
        if input.dim() < 2:
            raise ValueError("FocalLoss: logits need a class dimension (b, ch, ...)")
        return LossFn.apply(input, target, float(self.gamma), self.reduction == "mean", True, _weight_on(self.class_weight, input))


dict_losses = {
    "bce_loss": nn.BCELoss,
    "bce_wlogits_loss": nn.BCEWithLogitsLoss,
    "CrossEntropyLoss": CrossEntropyLoss,
    "FocalLoss": FocalLoss,
}
