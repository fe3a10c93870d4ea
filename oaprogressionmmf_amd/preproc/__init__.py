from ._pt import PTBatchAugment, PTInterpolate
