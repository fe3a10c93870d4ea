"""linear-layer shapes of the fusion transformers (GPU box)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in [(992, 2048, 2048), (992, 6144, 2048), (992, 2048, 6144), (520, 2048, 2048), (256, 2048, 2048), (208, 2048, 2048)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; dy = torch.randn(M, N, device=dev)
    b = torch.zeros(N, device=dev); fl = 2.0 * M * N * K
    f = fl / timeit(lambda: ops.linear_fwd(x, w, b, M, N, K)) / 1e9
    d = fl / timeit(lambda: ops.linear_dgrad(dy, w, M, N, K)) / 1e9
    print(f"M{M} N{N} K{K}: fwd {f:6.1f} dgrad {d:6.1f} TF/s")
