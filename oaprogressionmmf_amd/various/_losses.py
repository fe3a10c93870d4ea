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


class CrossEntropyLoss(nn.Module):
    def __init__(self, num_classes, batch_avg=True, batch_weight=None, class_avg=True, class_weight=None, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.batch_avg = batch_avg
        self.class_avg = class_avg
        self.batch_weight = batch_weight
        self.class_weight = class_weight
        logger.warning(f"Redundant loss function arguments:\n{repr(kwargs)}")
        if class_weight is not None:
            raise NotImplementedError("class_weight is not built (the reference recipes never set it)")

    def forward(self, input, target, **kwargs):
        return LossFn.apply(input, target, 0.0, True, False)


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
        if class_weight is not None:
            raise NotImplementedError("class_weight is not built (the reference recipes never set it)")

    def forward(self, input, target, **kwargs):
        """input (B, C) logits, target (B,) int64 -> scalar: mean|sum of -(1-pt)^gamma * log pt"""
        if input.dim() != 2:
            raise NotImplementedError("only (B, C) logits are built (the train loop passes (B, 2))")
        return LossFn.apply(input, target, float(self.gamma), self.reduction == "mean", True)


dict_losses = {
    "bce_loss": nn.BCELoss,
    "bce_wlogits_loss": nn.BCEWithLogitsLoss,
    "CrossEntropyLoss": CrossEntropyLoss,
    "FocalLoss": FocalLoss,
}
