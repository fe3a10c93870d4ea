"""microbench: 1x1 convolutions with the input from activation plane images against the fp32 loader"""
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
for (N_, H, W, Cin, Cout) in [(320, 96, 96, 64, 256), (320, 48, 48, 128, 512), (320, 24, 24, 256, 1024), (320, 12, 12, 512, 2048),
                              (320, 24, 24, 1024, 256), (320, 48, 48, 512, 128), (320, 12, 12, 2048, 512), (320, 96, 96, 256, 64)]:
    x = torch.randn(N_, H, W, Cin, device=dev); w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    fl = 2.0 * N_ * H * W * Cout * Cin
    img = ops.build_weight_planes(w, Cout, 1, Cin)
    out = []
    for ap in (False, True):
        t = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, 1, 1, 1, 0, sc, sh, stats=True, wimg=img, aplanes=ap))
        out.append(f"aplanes={int(ap)} {t:7.3f} ms {fl/t/1e9:6.1f} TF/s")
    t = timeit(lambda: ops.act_planes(x, N_ * H * W, Cin, 1, sc, sh, fscale=16.0))
    print(f"{Cin}->{Cout} @{H}: " + " | ".join(out) + f" | pre-pass {t:6.3f} ms", flush=True)
