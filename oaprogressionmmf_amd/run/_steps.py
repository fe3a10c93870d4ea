"""One train iteration / one inference batch (reference: koafusion/run/train_prog_fus.py:100-168,
koafusion/run/eval_prog_fus.py:262-303)."""
import torch

from .. import ops, preproc
from .._lib import KoafError


def downscale_inputs(xs, factors):
    """"Last-chance preprocessing" of both drivers (train_prog_fus.py:111-123, eval_prog_fus.py:245-276):
    factors is config.model.downscale -- falsy, or one scale tuple (or falsy) per modality."""
    if not factors:
        return tuple(xs)
    out = []
    for x, f in zip(xs, factors):
        if f:
            x = preproc.PTInterpolate(scale_factor=tuple(f))(x).contiguous()
        out.append(x)
    return tuple(out)


def train_step(model, loss_fn, optimizer, xs, ys, downscale=None):
    """zero_grad -> forward -> loss -> backward -> (all-reduce) -> Adam, the optimize branch of
    train_prog_fus.py:132-168.  `model` may be a registry model or a DataParallelRCCL wrapper.
    Returns (logits, loss), both on the device; nothing here synchronises the host."""
    xs = downscale_inputs(xs, downscale)
    optimizer.zero_grad()
    logits = model(*xs)["main"]
    loss = loss_fn(logits.squeeze(1), ys.long().squeeze(1))
    scale = getattr(model, "scale_loss", None)
    (scale(loss) if scale is not None else loss).backward()
    reduce = getattr(model, "reduce_gradients", None)
    if reduce is not None:
        reduce()
    optimizer.step()
    return logits.detach(), loss.detach()


def train_epoch(model, loss_fn, optimizer, batches, downscale=None):
    """the optimize branch of `ProgressionPrediction.train_epoch` (train_prog_fus.py:132-168) over an iterable of (xs, ys) device
    batches: one train_step each, the losses read back once at the end (no per-step host sync), and ONE look at the numerics
    status words per epoch (ops.check_numerics: clamped activations / non-finite operand scales raise a RuntimeWarning).
    Returns the list of per-step losses (python floats)."""
    losses = []
    for xs, ys in batches:
        losses.append(train_step(model, loss_fn, optimizer, xs, ys, downscale)[1])
    out = [float(v) for v in torch.stack(losses).cpu()] if losses else []
    if losses:
        ops.check_numerics()
    return out


def softmax_rows_(x):
    """in-place row softmax of a contiguous fp32 (rows, n) device tensor (koaf_softmax_rows)"""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 2):
        raise KoafError("softmax_rows_: contiguous fp32 (rows, n) device tensor required")
    if x.shape[0]:
        ops.check(ops.lib().koaf_softmax_rows(x.data_ptr(), x.shape[0], x.shape[1], ops._stream()), "softmax_rows")
    return x


def predict_batch(model, xs, downscale=None):
    """Inference on one batch (eval_prog_fus.py:262-303).  Returns (logits, proba) device tensors of shape
    (B, classes); the model must already be in eval() mode, autograd is off inside."""
    with torch.no_grad():
        xs = downscale_inputs(xs, downscale)
        logits = model(*xs)["main"]
        logits = logits.contiguous()      # (B, head*cls), "b head cls -> b (head cls)" in every model
        proba = softmax_rows_(logits.clone())
    return logits, proba
