"""Evaluation regime of koafusion/run/eval_prog_fus.py on the MI355X path: per-fold inference accumulators
(eval_epoch, :249-315), fold ensembling (ensemble_eval_foldw, :317-343) and the profile="time" harness
(:291-313) with the device actually drained around the timed region."""
import time
from collections import defaultdict

import numpy as np
import torch

from ._steps import predict_batch


class InferenceTimer(object):
    """Wall time of the model call per sample.  The reference brackets the asynchronous launch with
    time.time() (eval_prog_fus.py:291-297) and therefore under-reports on a GPU; here both edges
    synchronise the device, so the figure is launch + execution."""

    def __init__(self):
        self.sum_time = 0.0
        self.sum_samples = 0

    def __enter__(self):
        torch.cuda.synchronize()
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        self.sum_time += time.perf_counter() - self._t0
        return False

    def add_samples(self, n):
        self.sum_samples += int(n)

    @property
    def per_sample(self):
        return self.sum_time / self.sum_samples if self.sum_samples else float("nan")


def _extract_modal(batch, modal):
    """eval_prog_fus.py:240-243"""
    if modal not in ("sag_3d_dess", "cor_iw_tse", "sag_t2_map", "xr_pa", "clin"):
        raise AssertionError(f"unknown modality {modal!r}")
    return batch[f"image__{modal}"]


def eval_epoch(model, loader, modals, downscale=None, device="cuda", profile="none", timer=None):
    """One pass of an eval()-mode model over `loader` (batches are the reference's dicts: image__<modal>,
    target, ("-", "exam_knee_id")).  Returns the reference's accumulator dict (python lists, loader order):
    exam_knee_id, target, predict, predict_proba.  profile="time" fills `timer` (an InferenceTimer) around
    the model call only, as the reference does."""
    if profile not in ("none", "time"):
        raise ValueError(f"profile {profile!r}: only 'none' and 'time' are built (thop MAC counting is not)")
    if profile == "time" and timer is None:
        timer = InferenceTimer()
    acc = defaultdict(list)
    for batch in loader:
        xs = tuple(_extract_modal(batch, m).to(device) for m in modals)
        ys = batch["target"]
        if profile == "time":
            with timer:
                logits, proba = predict_batch(model, xs, downscale)
            timer.add_samples(xs[0].shape[0])
        else:
            logits, proba = predict_batch(model, xs, downscale)
        lg = logits.to("cpu")
        acc["exam_knee_id"].extend(batch[("-", "exam_knee_id")])
        acc["target"].extend(torch.as_tensor(ys).to("cpu").numpy().tolist())
        acc["predict"].extend(torch.argmax(lg, dim=1).numpy().tolist())
        acc["predict_proba"].extend(proba.to("cpu").tolist())
    if acc:
        from .. import ops
        ops.check_numerics()     # once per pass: clamped activations / non-finite operand scales warn instead of staying silent
    return dict(acc)


def ensemble_eval_foldw(raw_foldw):
    """Merge the per-fold accumulators on exam_knee_id (1:1, inner, first fold's order) and average:
    predict_proba = softmax(mean_folds(predict_proba__k)) -- softmax over probabilities, exactly the
    reference's arithmetic (eval_prog_fus.py:334-339, float64) -- predict = argmax."""
    folds = list(raw_foldw)
    if not folds:
        raise ValueError("no folds to ensemble")
    pos = {}
    for k in folds:
        ids = raw_foldw[k]["exam_knee_id"]
        pos[k] = dict(zip(ids, range(len(ids))))
        if len(pos[k]) != len(ids):
            raise ValueError(f"fold {k}: exam_knee_id values are not unique (1:1 merge)")
    k0 = folds[0]
    common = set(pos[k0]).intersection(*(pos[k].keys() for k in folds[1:]))
    ids = [e for e in raw_foldw[k0]["exam_knee_id"] if e in common]
    ens = {"exam_knee_id": ids, "target": [raw_foldw[k0]["target"][pos[k0][e]] for e in ids]}
    if not ids:
        for k in folds:
            ens[f"predict__{k}"], ens[f"predict_proba__{k}"] = [], []
        ens["predict_proba"], ens["predict"] = [], []
        return ens
    stack = []
    for k in folds:
        rows = [pos[k][e] for e in ids]
        ens[f"predict__{k}"] = [raw_foldw[k]["predict"][r] for r in rows]
        ens[f"predict_proba__{k}"] = [raw_foldw[k]["predict_proba"][r] for r in rows]
        stack.append(np.asarray(ens[f"predict_proba__{k}"], dtype=np.float64).reshape(len(ids), -1))
    mean = np.mean(np.stack(stack, axis=1), axis=1)                  # samples x classes
    e = np.exp(mean - mean.max(axis=-1, keepdims=True))
    proba = e / e.sum(axis=-1, keepdims=True)
    ens["predict_proba"] = proba.tolist()
    ens["predict"] = np.argmax(proba, axis=-1).tolist()
    return ens
