"""accuracy of the GEMM vs float64 on random data (GPU box): relative L2 error and max abs error / rms"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for M, N, K in [(512, 512, 512), (1024, 256, 4608), (2048, 2048, 2048)]:
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g) * 2)   # wide dynamic range
    w = torch.randn(N, K, generator=g) * K ** -0.5
    ref = x.double() @ w.double().t()
    y = ops.linear_fwd(x.to(dev), w.to(dev), None, M, N, K).cpu().double()
    y32 = (x.to(dev) @ w.to(dev).t()).cpu().double()
    rel = lambda a: ((a - ref).norm() / ref.norm()).item()
    print(f"M{M} N{N} K{K}: koaf rel {rel(y):.3e}  torch-fp32(rocBLAS) rel {rel(y32):.3e}  max|d|/rms koaf {((y-ref).abs().max()/ref.pow(2).mean().sqrt()).item():.3e}")
