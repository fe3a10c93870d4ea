"""microbench of the halo / gather plane-image kernels alone (images cut beforehand)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N_, H, W, Cin, Cout) in [(320, 96, 96, 64, 64), (320, 48, 48, 128, 128), (320, 24, 24, 256, 256), (320, 12, 12, 512, 512)]:
    k, s, p = 3, 1, 1
    x = torch.randn(N_, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    fl = 2.0 * N_ * H * W * Cout * 9 * Cin
    img = ops.build_weight_planes(w, Cout, 9, Cin)
    pl = ops.act_planes(x, N_ * H * W, Cin, 1, sc, sh, fscale=16.0)
    out = []
    for halo, name in ((0, "gather"), (2, "halo256"), (3, "halo128")):
        ops.set_conv3x3_halo(halo)
        t = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=img, aplanes=pl))
        out.append(f"{name} {t:7.3f} ms {fl/t/1e9:6.1f} TF/s")
    print(f"{Cin}->{Cout} @{H}: " + " | ".join(out), flush=True)
