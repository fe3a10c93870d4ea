"""microbench: the expanding 1x1 convolutions (N >= 4 K: every A tile is converted once per 128-column tile by the fp32 loader) with
the input from activation plane images against the fp32 loader; the pre-pass is timed beside the kernel"""
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
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


NS = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
shapes = [(NS, 96, 64, 256), (NS, 48, 128, 512), (NS, 24, 256, 1024), (NS, 12, 512, 2048), (NS, 24, 1024, 256), (NS, 48, 512, 128)]
for (N_, H, Cin, Cout) in shapes:
    x = torch.randn(N_, H, H, Cin, device=dev); w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    fl = 2.0 * N_ * H * H * Cout * Cin
    by = 4.0 * N_ * H * H * (Cin + Cout)
    img = ops.build_weight_planes(w, Cout, 1, Cin)
    t0 = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, H, Cin, Cout, 1, 1, 1, 0, sc, sh, stats=True, wimg=img, aplanes=False))
    tp = timeit(lambda: ops.act_planes(x, N_ * H * H, Cin, 1, sc, sh, fscale=16.0))
    pl = ops.act_planes(x, N_ * H * H, Cin, 1, sc, sh, fscale=16.0)
    t1 = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, H, Cin, Cout, 1, 1, 1, 0, sc, sh, stats=True, wimg=img, aplanes=pl))
    y0, p0 = ops.conv2d_fwd(x, w, N_, H, H, Cin, Cout, 1, 1, 1, 0, sc, sh, stats=True, wimg=img, aplanes=False)
    y1, p1 = ops.conv2d_fwd(x, w, N_, H, H, Cin, Cout, 1, 1, 1, 0, sc, sh, stats=True, wimg=img, aplanes=pl)
    same = bool(torch.equal(y0, y1))
    print(f"conv fwd 1x1 {Cin}->{Cout} px{N_*H*H}: fp32 loader {t0:7.3f} ms ({by/t0/1e9:5.2f} TB/s {fl/t0/1e9:6.1f} TF/s) | planes kernel {t1:7.3f} ms"
          f" + pre-pass {tp:6.3f} ms = {t1+tp:7.3f} | bit-identical {same}", flush=True)
    del x, y0, y1, pl
