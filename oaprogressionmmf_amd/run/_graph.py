"""HIP-graph replay of the inference pass.

One patient at a time the forward of the fusion model is launch-bound (about 1 700 kernel launches for four encoders and
four transformers against a few milliseconds of GPU work): capturing `predict_batch` once into a HIP graph and replaying it
removes the per-launch host cost (batch 1, native sizes: 13.2 -> 7.9 ms).  The encoder lanes (one HIP stream each) fork
from and join the capturing stream inside the capture, so the graph keeps their concurrency.

Only the evaluation regime is captured: training draws dropout seeds on the host every step and updates BatchNorm
running statistics through host-visible counters, which a static graph would freeze."""
import torch

from ._steps import predict_batch


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
