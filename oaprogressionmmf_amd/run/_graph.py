"""HIP-graph replay of the inference pass and of the train step.

One patient at a time the forward of the fusion model is launch-bound (about 1 700 kernel launches for four encoders and
four transformers against a few milliseconds of GPU work): capturing `predict_batch` once into a HIP graph and replaying it
removes the per-launch host cost (batch 1, native sizes: 13.2 -> 7.9 ms).  The encoder lanes (one HIP stream each) fork
from and join the capturing stream inside the capture, so the graph keeps their concurrency.

The TRAIN step (train_prog_fus.py:132-168) is captured too (GraphedTrainStep): everything that changes from step to step
lives in device memory -- the dropout step counter folded into every mask seed (functional.DeviceStepState), Adam's update
count and learning rate (Adam(capturable=True), koaf_adam_hyper), BatchNorm's num_batches_tracked (bumped by the finalize
kernel), the weight plane images (rebuilt by a node of the graph) -- and the lane joins at the end of backward() are
recorded inside the capture."""
import torch

from .. import functional as KF
from ._steps import predict_batch, train_step


class GraphedPredictor(object):
    """predict = GraphedPredictor(model, example_inputs[, downscale]);  logits, proba = predict(*inputs)

    `model` must be in eval() mode and stay unchanged in structure; inputs must keep the example's shapes and dtype
    (new values are copied into the captured input buffers).  The returned tensors are owned by the graph and are
    overwritten by the next call -- clone them to keep them."""

    def __init__(self, model, example_inputs, downscale=None, warmup=2):
        if model.training:
            raise RuntimeError("GraphedPredictor captures the evaluation regime: call model.eval() first")
        self.model, self.downscale = model, downscale
        self.static_in = [x.detach().clone().contiguous() for x in example_inputs]
        if not all(x.is_cuda for x in self.static_in):
            raise RuntimeError("GraphedPredictor needs inputs on the HIP device (there is no CPU path)")
        side = torch.cuda.Stream(device=self.static_in[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the capture: arena adoption, lazy workspaces
            for _ in range(max(1, warmup)):
                predict_batch(self.model, self.static_in, self.downscale)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = predict_batch(self.model, self.static_in, self.downscale)

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_in):
            raise TypeError(f"expected {len(self.static_in)} inputs, got {len(inputs)}")
        for dst, src in zip(self.static_in, inputs):
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"input shape {tuple(src.shape)} differs from the captured {tuple(dst.shape)}")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_out


class GraphedTrainStep(object):
    """step = GraphedTrainStep(model, loss_fn, optimizer, example_xs, example_ys[, downscale]);  logits, loss = step(xs, ys)

    The first `warmup` calls run the step eagerly (arena adoption, optimizer state, allocator warm-up: real training steps),
    the next call captures it into a HIP graph and replays it, later calls only copy the batch into the captured input
    buffers and replay.  Eager and replayed steps are the same kernels on the same device-resident step state, so a run
    is bit-identical whether or not (and when) it switches to replay.  `optimizer` must be Adam/AdamW(capturable=True);
    `model` a registry model on one GPU (the RCCL exchange of DataParallelRCCL is not captured).  The returned tensors are
    owned by the graph and overwritten by the next call.  Learning-rate changes (schedulers) are picked up at every call."""

    def __init__(self, model, loss_fn, optimizer, example_xs, example_ys, downscale=None, warmup=2, seed=None):
        if not getattr(optimizer, "capturable", False):
            raise RuntimeError("GraphedTrainStep needs Adam/AdamW(capturable=True): step count and learning rate on the device")
        if hasattr(model, "reduce_gradients"):
            raise RuntimeError("GraphedTrainStep captures single-GPU steps (the RCCL gradient exchange is not captured)")
        self.model, self.loss_fn, self.opt, self.downscale = model, loss_fn, optimizer, downscale
        self.xs = [x.detach().clone().contiguous() for x in example_xs]
        self.ys = example_ys.detach().clone().contiguous()
        if not all(x.is_cuda for x in self.xs):
            raise RuntimeError("GraphedTrainStep needs inputs on the HIP device (there is no CPU path)")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())    # (set_ultimate_seed makes it repeatable)
        self.state = KF.DeviceStepState(self.xs[0].device, seed)
        self.warmup, self.calls, self.graph, self.out = int(warmup), 0, None, None

    def _body(self):
        # (the per-lane weight-gradient side streams stay off in these steps: with encoder lanes AND side streams forked
        # inside one capture, ROCm 7.2's hipStreamEndCapture crashes; each alone captures fine.  Same kernels, same bits --
        # only the wgrad / BatchNorm-backward overlap inside a lane is given up, which launch-bound steps do not miss.)
        from ..models import _encoder
        prev, KF.STEP_STATE = KF.STEP_STATE, self.state
        side, _encoder.USE_SIDE_STREAM = _encoder.USE_SIDE_STREAM, False
        try:
            self.state.begin_step()
            return train_step(self.model, self.loss_fn, self.opt, self.xs, self.ys, self.downscale)
        finally:
            KF.STEP_STATE = prev
            _encoder.USE_SIDE_STREAM = side

    def __call__(self, xs, ys):
        if len(xs) != len(self.xs):
            raise TypeError(f"expected {len(self.xs)} inputs, got {len(xs)}")
        for dst, src in zip(self.xs + [self.ys], list(xs) + [ys]):
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"input shape {tuple(src.shape)} differs from the captured {tuple(dst.shape)}")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup:
            return self._body()
        self.opt.sync_hyper()
        if self.graph is None:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._body()
        self.graph.replay()
        self._invalidate_planes()
        return self.out

    def _invalidate_planes(self):
        """A replayed optimizer step rewrites the arena weights on the device without the host noticing (`arena.epoch` is only
        bumped while the step is being captured): move the epoch on after every replay, so that the next EAGER forward
        (validation between replayed train steps) finds its plane-image stamp stale and rebuilds the images from the
        weights this replay left, instead of multiplying with the ones cut at the start of the replay."""
        seen = set()
        for group in self.opt.param_groups:
            for p in group["params"]:
                a = getattr(p, "_koaf_arena", None)
                if a is not None and id(a) not in seen:
                    seen.add(id(a))
                    a.epoch += 1
