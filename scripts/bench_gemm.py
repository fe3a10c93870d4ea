"""microbench: the koaf GEMM on plain dense problems (all four operand-kind pairs) + a few conv shapes"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in [(4096, 4096, 4096), (8192, 8192, 2048), (16384, 2048, 2048), (51200, 256, 2304), (204800, 128, 1152)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; dy = torch.randn(M, N, device=dev)
    dw = torch.empty(N, K, device=dev)
    fl = 2.0 * M * N * K
    t = timeit(lambda: ops.linear_fwd(x, w, None, M, N, K)); print(f"fwd   KC/KC M{M} N{N} K{K}: {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.linear_dgrad(dy, w, M, N, K)); print(f"dgrad KC/KM M{M} N{N} K{K}: {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.linear_wgrad(dy, x, dw, None, M, N, K)); print(f"wgrad KM/KM M{M} N{N} K{K}: {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")

shapes = [(512, 10, 10, 256, 256, 3, 1, 1), (512, 20, 20, 128, 128, 3, 1, 1), (512, 40, 40, 64, 64, 3, 1, 1), (512, 10, 10, 256, 1024, 1, 1, 0),
          # the synthetic-shape (384^2 slices) layers, 160 slices
          (160, 24, 24, 256, 256, 3, 1, 1), (160, 96, 96, 64, 64, 3, 1, 1), (160, 96, 96, 64, 256, 1, 1, 0), (160, 96, 96, 256, 64, 1, 1, 0),
          (160, 48, 48, 128, 512, 1, 1, 0), (160, 24, 24, 1024, 256, 1, 1, 0)]
for (N_, H, W, Cin, Cout, k, s, p) in shapes:
    x = torch.randn(N_, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    OH = ops.conv_out(H, k, s, p); fl = 2.0 * N_ * OH * OH * Cout * k * k * Cin
    dy = torch.randn(N_, OH, OH, Cout, device=dev); dw = torch.empty_like(w)
    t = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True)); print(f"conv fwd  k{k} {Cin}->{Cout} px{N_*OH*OH}: {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    img = ops.build_weight_planes(w, Cout, k * k, Cin)
    am = dy.abs().max().reshape(1); noimg = (None, None, img[2])
    t = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=noimg)); print(f"conv fwd  f16 (weights cut in-kernel): {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=img)); print(f"conv fwd  f16 + weight images      : {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, s, p, None, None, stats=False)); print(f"conv fwd(no tf/stats)             : {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.conv2d_dgrad(dy, w, N_, H, W, Cin, Cout, k, k, s, p)); print(f"conv dgrad                        : {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.conv2d_dgrad(dy, w, N_, H, W, Cin, Cout, k, k, s, p, wimg=noimg, dy_amax=am)); print(f"conv dgrad f16 (in-kernel)         : {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.conv2d_dgrad(dy, w, N_, H, W, Cin, Cout, k, k, s, p, wimg=img, dy_amax=am)); print(f"conv dgrad f16 + weight images     : {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.conv2d_wgrad(dy, x, dw, N_, H, W, Cin, Cout, k, k, s, p, sc, sh)); print(f"conv wgrad                        : {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.conv2d_wgrad(dy, x, dw, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, dy_amax=am)); print(f"conv wgrad f16                    : {t:8.3f} ms {fl/t/1e9:7.1f} TF/s")
