"""Model registry -- same keys and constructor protocol as koafusion/models/__init__.py:8-15:
`dict_models[config.model.name](config=config.model, path_weights=...)`."""
from ._core_fes import dict_fes
from ._core_trf import Attention, FeaT, FeedForward, Transformer
from ._encoder import KoafTrunk
from ._mrN_cnn_trf import MR1CnnTrf, MR2CnnTrf
from ._xr1_cnn import XR1Cnn
from ._xr1mrN import XR1MR1CnnTrf, XR1MR2CnnTrf
from ._xrNmrMcP import FeatC1, XR1MR2C1CnnTrf

dict_models = {
    "XR1Cnn": XR1Cnn,
    "MR1CnnTrf": MR1CnnTrf,
    "MR2CnnTrf": MR2CnnTrf,
    "XR1MR1CnnTrf": XR1MR1CnnTrf,
    "XR1MR2CnnTrf": XR1MR2CnnTrf,
    "XR1MR2C1CnnTrf": XR1MR2C1CnnTrf,
}
