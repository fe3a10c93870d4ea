import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, H, W, Cin, Cout, k, s, p = 13, 64, 64, 64, 128, 1, 1, 0
x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.1
img = ops.build_weight_planes(w, Cout, 1, Cin)
noimg = (None, None, img[2])
a = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, k, k, s, p, None, None, stats=True, wimg=noimg)
b = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, k, k, s, p, None, None, stats=True, wimg=img, aplanes=False)
d = (a[0] - b[0]).abs().reshape(-1, Cout)
print("out max diff", float(d.max()), "rows bad", int((d.amax(1) > 0).sum()), "of", d.shape[0])
bad = (d.amax(1) > 0).nonzero().flatten()
print("first bad rows", bad[:20].tolist(), "row%128", (bad[:20] % 128).tolist())
print("stats equal", torch.equal(a[1], b[1]), a[1].shape, b[1].shape)
ds = (a[1] - b[1]).abs()
print("stats max diff", float(ds.max()), "rel", float(ds.max() / a[1].abs().max()))
