"""microbench: weight gradients with both operands from activation plane images (K-major LDS-DMA) against the fp32 loaders;
the plane-image times include cutting x (dy's images exist already: the data gradient cut them)"""
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
shapes = [(320, 96, 96, 64, 64, 3, 1), (320, 48, 48, 128, 128, 3, 1), (320, 24, 24, 256, 256, 3, 1), (320, 12, 12, 512, 512, 3, 1),
          (320, 96, 96, 128, 128, 3, 2), (320, 96, 96, 64, 256, 1, 1), (320, 96, 96, 256, 64, 1, 1), (320, 24, 24, 256, 1024, 1, 1),
          (320, 24, 24, 1024, 256, 1, 1)]
for (N_, H, W, Cin, Cout, k, s) in shapes:
    p = k // 2
    OH = ops.conv_out(H, k, s, p)
    x = torch.randn(N_, H, W, Cin, device=dev); dy = torch.randn(N_, OH, OH, Cout, device=dev) * 1e-3
    dw = torch.empty(Cout, k, k, Cin, device=dev)
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    am = dy.abs().max().reshape(1)
    fl = 2.0 * N_ * OH * OH * Cout * k * k * Cin
    out = []
    for ap in (False, True):
        t = timeit(lambda: ops.conv2d_wgrad(dy, x, dw, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, dy_amax=am, aplanes=ap))
        out.append(f"aplanes={int(ap)} {t:7.3f} ms {fl/t/1e9:6.1f} TF/s")
    t1 = timeit(lambda: ops.act_planes(x, N_ * H * W, Cin, 1, sc, sh, fscale=16.0))
    t2 = timeit(lambda: ops.act_planes(dy, N_ * OH * OH, Cout, 0, amax=am))
    print(f"wgrad k{k}s{s} {Cin}->{Cout} @{H}: " + " | ".join(out) + f" | cut x {t1:6.3f} dy {t2:6.3f} ms", flush=True)
