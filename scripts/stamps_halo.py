"""In-kernel phase stamps of the 3x3 halo kernels (diagnostic build: make -C oaprogressionmmf_amd/csrc stamps;
KOAF_LIB=oaprogressionmmf_amd/csrc/libkoaf_stamps.so python scripts/stamps_halo.py): per tile, the average time from block
entry to the first k-step, in the k-loop (of which exposed waits at channel-chunk switches), staging the accumulators,
stores + statistics; and the kernel's wall time."""
import ctypes
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
from oaprogressionmmf_amd._lib import lib
dev = torch.device("cuda:0")
L = lib()
buf = (ctypes.c_ulonglong * 8)()


def stamps(reset):
    L.koaf_debug_stamps(buf, 1 if reset else 0)
    return list(buf)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    stamps(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, stamps(False)


shapes = [(1280, 96, 96, 64, 64), (1280, 48, 48, 128, 128), (1280, 24, 24, 256, 256), (1280, 12, 12, 512, 512)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (N_, H, W, Cin, Cout) in shapes:
    k, s, p = 3, 1, 1
    x = torch.randn(N_, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    fl = 2.0 * N_ * H * W * Cout * 9 * Cin
    img = ops.build_weight_planes(w, Cout, 9, Cin)
    pl = ops.act_planes(x, N_ * H * W, Cin, 1, sc, sh, fscale=16.0)
    for halo, name in ((2, "halo256"), (3, "halo128"), (1, "auto (2-D tiles for 64 channels)")):
        ops.set_conv3x3_halo(halo)
        for st in (True, False):
            t, b = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=st, wimg=img, aplanes=pl))
            n = max(b[7], 1)
            us = [v / n / 100.0 for v in b[:7]]
            print(f"{Cin}->{Cout} @{H} {name} stats={int(st)}: {t:7.3f} ms {fl/t/1e9:6.1f} TF/s | tiles/launch {b[7] // 5}: per tile us: "
                  f"prologue {us[0]:.2f} k-loop {us[1]:.2f} (counted waits {us[5]:.2f}, barriers {us[6]:.2f}) stage {us[2]:.2f} store+stats {us[3]:.2f} total {us[4]:.2f}",
                  flush=True)
