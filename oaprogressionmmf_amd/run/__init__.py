"""Step bodies of the reference's drivers (koafusion/run/train_prog_fus.py, eval_prog_fus.py) on the MI355X
path.  The drivers themselves (hydra, data loaders, tensorboard, metrics) stay the reference's."""
from ._steps import downscale_inputs, train_epoch, train_step, predict_batch
from ._eval import eval_epoch, ensemble_eval_foldw, InferenceTimer
from ._graph import GraphedPredictor, GraphedTrainStep
from ._explain import explain_epoch, ensemble_explain_foldw, modal_ablation, ablation_percent

__all__ = ["downscale_inputs", "train_epoch", "train_step", "predict_batch", "eval_epoch", "ensemble_eval_foldw",
           "InferenceTimer", "GraphedPredictor", "GraphedTrainStep", "explain_epoch", "ensemble_explain_foldw", "modal_ablation",
           "ablation_percent"]
