import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (N_, H, W, Cin, Cout, k, s) in [(1, 5, 5, 512, 512, 3, 1), (1, 5, 5, 64, 64, 3, 1), (1, 6, 6, 64, 64, 3, 1), (2, 4, 4, 128, 64, 3, 1), (1, 3, 3, 64, 64, 3, 1)]:
    p = 1
    x = torch.randn(N_, H, W, Cin, device=dev); dy = torch.randn(N_, H, W, Cout, device=dev) * 1e-3
    am = dy.abs().max().reshape(1)
    a0, a1 = torch.empty(Cout, k, k, Cin, device=dev), torch.empty(Cout, k, k, Cin, device=dev)
    ops.conv2d_wgrad(dy, x, a0, N_, H, W, Cin, Cout, k, k, s, p, dy_amax=am, aplanes=False)
    ops.conv2d_wgrad(dy, x, a1, N_, H, W, Cin, Cout, k, k, s, p, dy_amax=am, aplanes=True)
    d = (a0 - a1).abs()
    print((N_, H, W, Cin, Cout), "max diff", float(d.max()), "ref max", float(a0.abs().max()), "bad taps", (d.amax(dim=(0, 3)) > 0).int().flatten().tolist())
