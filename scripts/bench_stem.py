"""microbench: the 7x7 / stride-2 stem (forward, weight gradient) on one encoder's slices of the synthetic shape"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


N, H, W = 1280, 384, 384
x = torch.randn(N, H, W, device=dev); w1t = torch.randn(49, 64, device=dev) * 0.1
y = ops.stem_fwd(x, w1t, N, H, W)
dy = torch.randn_like(y); dw = torch.empty(64, 7, 7, 3, device=dev)
fl = 2.0 * y.numel() * 49
t = timeit(lambda: ops.stem_fwd(x, w1t, N, H, W))
print(f"stem fwd  {t:7.3f} ms  {fl / t / 1e9:6.1f} TFLOP/s  {(y.numel() + x.numel()) * 4 / t / 1e9:5.2f} TB/s")
t = timeit(lambda: ops.stem_fwd(x, w1t, N, H, W, dtype=torch.bfloat16))
print(f"stem fwd (bf16 out) {t:7.3f} ms")
t = timeit(lambda: ops.stem_wgrad(dy, x, dw, N, H, W))
print(f"stem wgrad {t:7.3f} ms  {fl / t / 1e9:6.1f} TFLOP/s  {(y.numel() + x.numel()) * 4 / t / 1e9:5.2f} TB/s")
# the stem's backward chain as the trunk runs it: statistics in the forward kernel, the max-pool's input gradient gathered by
# the BatchNorm reduction, dc formed by the weight gradient on load
t = timeit(lambda: ops.stem_fwd(x, w1t, N, H, W, stats=True))
print(f"stem fwd + statistics {t:7.3f} ms")
H1, W1, C = y.shape[1], y.shape[2], 64
rows = N * H1 * W1
saved = ops.bn_finalize(ops.colstats(y, rows, C), C, rows, torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev),
                        torch.ones(C, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), 0.1, 1e-5, True)
yp, am = ops.maxpool_fwd(y, saved, N, H1, W1, C)
dyp = torch.randn_like(yp)
dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
dz = torch.empty_like(y)
t = timeit(lambda: ops.bn_bwd(None, y, saved, rows, C, rows, dg, db, 2, fused=True, pool=(dyp, am, N, H1, W1), dz_out=dz))
print(f"BatchNorm reduction with the pool gradient gathered {t:7.3f} ms  {(2 * y.numel() * 4 + dyp.numel() * 5) / t / 1e9:5.2f} TB/s")
ap = ops.bn_bwd(None, y, saved, rows, C, rows, dg, db, 2, fused=True, pool=(dyp, am, N, H1, W1), dz_out=dz)
t = timeit(lambda: ops.stem_wgrad(ap, x, dw, N, H, W))
print(f"stem wgrad with the apply on load {t:7.3f} ms")
