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
for (N_, H, W, C) in [(320, 96, 96, 128), (320, 48, 48, 256), (320, 24, 24, 512)]:
    k, s, p = 3, 2, 1
    OH = ops.conv_out(H, k, s, p)
    w = torch.randn(C, k, k, C, device=dev) * 0.05
    dy = torch.randn(N_, OH, OH, C, device=dev) * 1e-3
    am = dy.abs().max().reshape(1)
    img = ops.build_weight_planes(w, C, 9, C)
    fl = 2.0 * N_ * OH * OH * C * 9 * C
    out = []
    for ap in (False, True):
        t = timeit(lambda: ops.conv2d_dgrad(dy, w, N_, H, W, C, C, k, k, s, p, wimg=img, dy_amax=am, aplanes=ap))
        out.append(f"aplanes={int(ap)} {t:7.3f} ms {fl/t/1e9:6.1f} TF/s")
    print(f"dgrad k3s2 {C} @{H}: " + " | ".join(out), flush=True)
