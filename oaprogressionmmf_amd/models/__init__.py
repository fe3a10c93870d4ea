"""Model registry -- same keys and constructor protocol as koafusion/models/__init__.py:8-15:
`dict_models[config.model.name](config=config.model, path_weights=...)`."""
from ._core_fes import dict_fes
from ._core_trf import Attention, FeaT, FeedForward, Transformer
from ._encoder import KoafTrunk
from ._mrN_cnn_trf import MR1CnnTrf, MR2CnnTrf
from ._xr1_cnn import XR1Cnn
from ._xr1mrN import XR1MR1CnnTrf, XR1MR2CnnTrf
from ._xrNmrMcP import FeatC1, XR1MR2C1CnnTrf
from ._ext import MR1C1CnnTrf, XR1C1Cnn, XR1MR3C1CnnTrf

dict_models = {
    "XR1Cnn": XR1Cnn,
    "MR1CnnTrf": MR1CnnTrf,
    "MR2CnnTrf": MR2CnnTrf,
    "XR1MR1CnnTrf": XR1MR1CnnTrf,
    "XR1MR2CnnTrf": XR1MR2CnnTrf,
    "XR1MR2C1CnnTrf": XR1MR2C1CnnTrf,
}
REFERENCE_MODELS = tuple(dict_models)        # the reference's registry keys (koafusion/models/__init__.py:8-15)

# extensions for the BASELINE.json configurations without a reference class (see _ext.py)
dict_models.update({
    "XR1C1Cnn": XR1C1Cnn,
    "MR1C1CnnTrf": MR1C1CnnTrf,
    "XR1MR3C1CnnTrf": XR1MR3C1CnnTrf,
})
