"""Explanation regime of koafusion/run/eval_prog_fus.py (explain_epoch :410-479, ensemble_explain_foldw :481-512):
modality ablation.  The reference drives captum's FeatureAblation with one feature id per input tensor, no
baselines (= zeros) and one perturbation per evaluation; for that call captum's rule reduces to

    attr[b, m] = f(x)[b, target_b] - f(x with modality m zeroed)[b, target_b]

replicated over every element of input m (so the reference's mean over elements returns the difference itself).
Here the M + 1 forwards run on the HIP path and only the (B, M) differences leave the device."""
from collections import defaultdict

import numpy as np
import torch

from ._eval import _extract_modal
from ._steps import downscale_inputs


def _forward_main(model, xs):
    out = model(*xs)
    out = out["main"] if isinstance(out, dict) else out            # output_type "dict" / "main" (the captum path)
    return out.reshape(out.shape[0], -1)


def modal_ablation(model, xs, target):
    """(B, M) fp32 device tensor of the attributions above; `target` (B,) or (B, 1) integer class per sample."""
    tgt = torch.as_tensor(target).to(xs[0].device).long().reshape(-1, 1)
    if tgt.shape[0] == 1 and xs[0].shape[0] > 1:                   # a squeezed single target applies to every row
        tgt = tgt.expand(xs[0].shape[0], 1)
    with torch.no_grad():
        base = _forward_main(model, xs).gather(1, tgt)
        cols = []
        for m in range(len(xs)):
            ablated = tuple(torch.zeros_like(x) if j == m else x for j, x in enumerate(xs))
            cols.append(base - _forward_main(model, ablated).gather(1, tgt))
    return torch.cat(cols, dim=1)


def ablation_percent(attrs):
    """eval_prog_fus.py:456-459: rows normalised to unit L1, absolute value, per cent rounded to 3 decimals
    (fp32 arithmetic on the CPU copy, as there)."""
    t = torch.as_tensor(attrs, dtype=torch.float32).to("cpu")
    t = t / torch.sum(torch.abs(t), dim=1, keepdim=True)
    return np.round(np.abs(t.numpy()) * 100., decimals=3)


def explain_epoch(model, loader, modals, downscale=None, device="cuda", explain_fn="modal_abl"):
    """One pass of an eval()-mode model over `loader`; returns the reference's accumulator dict: exam_knee_id,
    target, modal_names, modal_abl_attrs, modal_abl_percent (python lists, loader order)."""
    if explain_fn != "modal_abl":
        raise ValueError(f"Unknown explain_fn: {explain_fn}")
    acc = defaultdict(list)
    modals = list(modals)
    for batch in loader:
        xs = tuple(_extract_modal(batch, m).to(device) for m in modals)
        ys = torch.as_tensor(batch["target"])
        with torch.no_grad():
            xs = tuple(downscale_inputs(xs, downscale))
        attrs = modal_ablation(model, xs, ys.squeeze()).to("cpu")
        acc["exam_knee_id"].extend(batch[("-", "exam_knee_id")])
        acc["target"].extend(ys.to("cpu").numpy().tolist())
        acc["modal_names"].extend([modals, ] * attrs.shape[0])
        acc["modal_abl_attrs"].extend(attrs.numpy().tolist())
        acc["modal_abl_percent"].extend(ablation_percent(attrs).tolist())
    return dict(acc)


def ensemble_explain_foldw(raw_foldw):
    """Inner 1:1 merge of the folds on exam_knee_id (first fold's order; target / modal_names from the first fold),
    per-fold columns modal_abl_attrs__k / modal_abl_percent__k, and modal_abl_percent = fold mean of the per-fold
    per-cent rows renormalised to sum 1 (a fraction, as the reference leaves it; float64)."""
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
    ens = {"exam_knee_id": ids}
    for field in ("target", "modal_names"):
        ens[field] = [raw_foldw[k0][field][pos[k0][e]] for e in ids]
    stack = []
    for k in folds:
        rows = [pos[k][e] for e in ids]
        ens[f"modal_abl_attrs__{k}"] = [raw_foldw[k]["modal_abl_attrs"][r] for r in rows]
        ens[f"modal_abl_percent__{k}"] = [raw_foldw[k]["modal_abl_percent"][r] for r in rows]
        if ids:
            stack.append(np.asarray(ens[f"modal_abl_percent__{k}"], dtype=np.float64).reshape(len(ids), -1))
    if not ids:
        ens["modal_abl_percent"] = []
        return ens
    mean = np.mean(np.stack(stack, axis=1), axis=1)                  # samples x modals
    ens["modal_abl_percent"] = (mean / np.sum(mean, axis=1, keepdims=True)).tolist()
    return ens
