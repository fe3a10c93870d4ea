"""A/B timing of the step's conv shapes (run once per library: KOAF_LIB=<.so>); interleave runs on ONE box"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=12):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(os.environ.get("KOAF_LIB", "default"))
for (N_, H, Cin, Cout, k) in [(512, 10, 256, 256, 3), (512, 20, 128, 128, 3), (128, 80, 64, 64, 3), (512, 20, 512, 128, 1), (512, 10, 1024, 256, 1),
                              (512, 20, 128, 512, 1), (128, 80, 64, 256, 1)]:
    W = H; p = k // 2
    x = torch.randn(N_, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    fl = 2.0 * N_ * H * W * Cout * k * k * Cin
    dy = torch.randn(N_, H, W, Cout, device=dev); dw = torch.empty_like(w)
    tf = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, 1, p, sc, sh, stats=True))
    td = timeit(lambda: ops.conv2d_dgrad(dy, w, N_, H, W, Cin, Cout, k, k, 1, p))
    tw = timeit(lambda: ops.conv2d_wgrad(dy, x, dw, N_, H, W, Cin, Cout, k, k, 1, p, sc, sh))
    print(f"k{k} {Cin:4d}->{Cout:4d} px{N_*H*W:7d}: fwd {fl/tf/1e9:6.1f}  dgrad {fl/td/1e9:6.1f}  wgrad {fl/tw/1e9:6.1f} TF/s")
