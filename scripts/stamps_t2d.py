import ctypes, sys
sys.path.insert(0, '.')
import torch
from oaprogressionmmf_amd import ops
from oaprogressionmmf_amd._lib import lib
dev = torch.device("cuda:0"); L = lib(); buf = (ctypes.c_ulonglong * 8)()
def stamps(reset):
    L.koaf_debug_stamps(buf, 1 if reset else 0); return list(buf)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); stamps(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, stamps(False)
N_, H, W, Cin, Cout = 1280, 96, 96, 64, 64
x = torch.randn(N_, H, W, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
fl = 2.0 * N_ * H * W * Cout * 9 * Cin
img = ops.build_weight_planes(w, Cout, 9, Cin)
pl = ops.act_planes(x, N_ * H * W, Cin, 1, sc, sh, fscale=16.0)
for halo, name in ((3, "halo128"), (1, "t2d")):
    ops.set_conv3x3_halo(halo)
    for st in (True, False):
        t, b = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, 3, 3, 1, 1, sc, sh, stats=st, wimg=img, aplanes=pl))
        n = max(b[7], 1); us = [v / n / 100.0 for v in b[:6]]
        print(f"{name} stats={int(st)}: {t:7.3f} ms {fl/t/1e9:6.1f} TF/s | tiles {b[7]//5}: prologue {us[0]:.2f} k-loop {us[1]:.2f} stage {us[2]:.2f} store+stats {us[3]:.2f} total {us[4]:.2f}", flush=True)
