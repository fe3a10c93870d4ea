"""GEMM launches for `rocprofv3 --pmc` counter collection: the step's heaviest matrix-pipe-bound conv shape (3x3 256->256 over
51200 pixels: forward on three bf16 pieces, data and weight gradients on two) and a 4096^3 dense product, REP launches each
after a warm-up, random operands (zeros clock higher)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M, N, K = 4096, 4096, 4096
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
for _ in range(REP): ops.linear_fwd(x, w, None, M, N, K)
N_, H, W, Cin, Cout, k, s, p = 512, 10, 10, 256, 256, 3, 1, 1
xx = torch.randn(N_, H, W, Cin, device=dev); ww = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
dy = torch.randn(N_, H, W, Cout, device=dev); dw = torch.empty_like(ww)
sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
for _ in range(REP): ops.conv2d_fwd(xx, ww, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True)
for _ in range(REP): ops.conv2d_dgrad(dy, ww, N_, H, W, Cin, Cout, k, k, s, p)
for _ in range(REP): ops.conv2d_wgrad(dy, xx, dw, N_, H, W, Cin, Cout, k, k, s, p, sc, sh)
torch.cuda.synchronize()
