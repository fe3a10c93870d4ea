"""(GPU box) the 256-row halo kernel (3x3 / stride 1 / pad 1 over plane images, C >= 128) alone on the three layer shapes of the synthetic
trunk that use it, 320 slices; with the stamps build (make -C oaprogressionmmf_amd/csrc stamps;
KOAF_LIB=oaprogressionmmf_amd/csrc/libkoaf_stamps.so) also the per-tile phase times of wave 0
    python scripts/bench_halo256.py [slices=320] [iterations=10]"""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
from oaprogressionmmf_amd._lib import lib
dev = torch.device("cuda:0")
L = lib()
buf = (ctypes.c_ulonglong * 8)()
has = hasattr(L, "koaf_debug_stamps") and "stamps" in os.environ.get("KOAF_LIB", "")
NIT = int(sys.argv[2]) if len(sys.argv) > 2 else 10          # (hundreds: a second of sustained load per shape, as inside the step)


def timeit(fn, n=NIT):
    fn(); torch.cuda.synchronize()
    if has: L.koaf_debug_stamps(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    if has: L.koaf_debug_stamps(buf, 0)
    return e0.elapsed_time(e1) / n
ops.set_conv3x3_halo(int(os.environ.get("KOAF_BENCH_HALO", "2")))       # (2: the 256-row shape; 3: 128 rows, two blocks per CU)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 320        # (1280 = one call of the headline step: the plane images no longer fit the 256 MB cache)
for (N_, H, W, C) in [(NS, 48, 48, 128), (NS, 24, 24, 256), (NS, 12, 12, 512)]:
    x = torch.randn(N_, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    img = ops.build_weight_planes(w, C, 9, C)
    pl = ops.act_planes(x, N_ * H * W, C, 1, sc, sh, fscale=16.0)
    t = timeit(lambda: ops.conv2d_fwd(x, w, N_, H, W, C, C, 3, 3, 1, 1, sc, sh, stats=True, wimg=img, aplanes=pl))
    fl = 2.0 * N_ * H * W * C * 9 * C
    msg = f"halo-256 forward {C}->{C}@{H}: {t:.3f} ms {fl / t / 1e9:6.1f} TF/s"
    if has:
        n = max(buf[7], 1); us = [buf[i] / n / 100.0 for i in range(7)]
        steps = 9 * C // 32
        msg += f" | per tile us: prologue {us[0]:.2f} k-loop {us[1]:.2f} ({us[1]/steps:.3f}/step; step-boundary wait + barrier {us[5]:.2f}, rest {us[6]:.2f}) stage {us[2]:.2f} store {us[3]:.2f} total {us[4]:.2f}"
    print(msg, flush=True)
