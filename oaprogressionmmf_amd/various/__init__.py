"""Mirror of koafusion/various/__init__.py:1-11 for the hot-path pieces."""
from ._checkpoint import CheckpointHandler, load_train_state, save_train_state
from ._losses import dict_losses
from ._optimizers import dict_optimizers, dict_schedulers
from ._seed import set_ultimate_seed
