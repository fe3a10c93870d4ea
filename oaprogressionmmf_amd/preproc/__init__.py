from ._pt import PTInterpolate
