"""short GEMM microbench for A/B experiments (GPU box)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = []
for M, N, K in [(8192, 8192, 2048), (16384, 2048, 2048)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; dy = torch.randn(M, N, device=dev)
    dw = torch.empty(N, K, device=dev); fl = 2.0 * M * N * K
    out.append(fl / timeit(lambda: ops.linear_fwd(x, w, None, M, N, K)) / 1e9)
    out.append(fl / timeit(lambda: ops.linear_dgrad(dy, w, M, N, K)) / 1e9)
    out.append(fl / timeit(lambda: ops.linear_wgrad(dy, x, dw, None, M, N, K)) / 1e9)
for (N_, H, W, Cin, Cout, k, s, p) in [(512, 10, 10, 256, 256, 3, 1, 1), (512, 20, 20, 128, 128, 3, 1, 1), (512, 40, 40, 64, 64, 3, 1, 1)]:
    x = torch.randn(N_, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    OH = ops.conv_out(H, k, s, p); fl = 2.0 * N_ * OH * OH * Cout * k * k * Cin
    dy = torch.randn(N_, OH, OH, Cout, device=dev); dw = torch.empty_like(w)
    out.append(fl / timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True)) / 1e9)
    out.append(fl / timeit(lambda: ops.conv2d_dgrad(dy, w, N_, H, W, Cin, Cout, k, k, s, p)) / 1e9)
    out.append(fl / timeit(lambda: ops.conv2d_wgrad(dy, x, dw, N_, H, W, Cin, Cout, k, k, s, p, sc, sh)) / 1e9)
print(" ".join(f"{v:6.1f}" for v in out))
