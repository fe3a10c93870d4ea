"""a handful of GEMM launches for PMC collection (one launch per shape, after one warm-up each)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
M, N, K = 4096, 4096, 4096
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
for _ in range(2): ops.linear_fwd(x, w, None, M, N, K)
for (N_, H, W, Cin, Cout, k, s, p) in [(512, 10, 10, 256, 256, 3, 1, 1), (512, 40, 40, 64, 64, 3, 1, 1), (512, 10, 10, 1024, 256, 1, 1, 0)]:
    xx = torch.randn(N_, H, W, Cin, device=dev); ww = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    for _ in range(2): ops.conv2d_fwd(xx, ww, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True)
torch.cuda.synchronize()
