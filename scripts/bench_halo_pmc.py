"""3x3 convolution launches for `rocprofv3 --pmc` on the plane-image kernels: per-tap gather (halo off) and halo kernel,
64->64 at 96x96 (80 slices) and 256->256 at 24x24 (160 slices)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for (N_, H, W, Cin, Cout) in [(80, 96, 96, 64, 64), (160, 24, 24, 256, 256)]:
    k, s, p = 3, 1, 1
    xx = torch.randn(N_, H, W, Cin, device=dev); ww = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    img = ops.build_weight_planes(ww, Cout, k * k, Cin)
    for halo in (False, True):
        ops.set_conv3x3_halo(halo)
        for _ in range(REP): ops.conv2d_fwd(xx, ww, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=img, aplanes=True)
torch.cuda.synchronize()
