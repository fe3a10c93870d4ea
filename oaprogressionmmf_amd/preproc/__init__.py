from ._pt import PTBatchAugment, PTInterpolate, PinnedPrefetcher
