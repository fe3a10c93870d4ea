"""microbench: the 3x3 convolutions of the synthetic-shape trunk with the input from activation plane images (DMA) against
the fp32 loader (conversion in the k-loop); times include the koaf_act_planes pre-pass"""
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


shapes = [(320, 96, 96, 64, 64, 3, 1, 1), (320, 48, 48, 128, 128, 3, 1, 1), (320, 24, 24, 256, 256, 3, 1, 1), (320, 12, 12, 512, 512, 3, 1, 1),
          (320, 96, 96, 128, 128, 3, 2, 1), (320, 48, 48, 256, 256, 3, 2, 1)]
for (N_, H, W, Cin, Cout, k, s, p) in shapes:
    x = torch.randn(N_, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    OH = ops.conv_out(H, k, s, p); fl = 2.0 * N_ * OH * OH * Cout * k * k * Cin
    img = ops.build_weight_planes(w, Cout, k * k, Cin)
    for ap, halo in ((False, False), (True, False), (True, True)):
        ops.set_conv3x3_halo(halo)
        t = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=img, aplanes=ap))
        print(f"conv fwd k{k}s{s} {Cin}->{Cout} px{N_*OH*OH} aplanes={int(ap)} halo={int(halo)}: {t:8.3f} ms {fl/t/1e9:7.1f} TF/s", flush=True)
    t = timeit(lambda: ops.act_planes(x, N_ * H * W, Cin, 1, sc, sh, fscale=16.0))
    print(f"   act_planes pre-pass alone: {t:8.3f} ms  {N_*H*W*Cin*8/t/1e9:6.2f} TB/s", flush=True)
