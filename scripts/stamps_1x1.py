"""In-kernel phase stamps of the 1x1 convolutions of the synthetic-shape trunk (diagnostic build: make -C
oaprogressionmmf_amd/csrc stamps; KOAF_LIB=oaprogressionmmf_amd/csrc/libkoaf_stamps.so python scripts/stamps_1x1.py):
forward (BatchNorm prologue on load, statistics epilogue on / off) and the data gradient with the BatchNorm-backward apply
formed in the loader and the fused BatchNorm-backward reduction in the epilogue."""
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
REP = 5


def stamps(reset):
    L.koaf_debug_stamps(buf, 1 if reset else 0)
    return list(buf)


def timeit(fn, n=REP):
    fn(); torch.cuda.synchronize()
    stamps(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, stamps(False)


def show(tag, t, b, byt):
    n = max(b[7], 1)
    us = [v / n / 100.0 for v in b[:7]]
    print(f"{tag:46s} {t:7.3f} ms {byt / t / 1e9:5.2f} TB/s | tiles/launch {b[7] // REP:6d}: per tile us: prologue {us[0]:5.2f} k-loop {us[1]:6.2f} "
          f"(streamed kernel: A wait + transform {us[6]:5.2f}, B wait + barrier {us[5]:5.2f}) stage {us[2]:5.2f} store+red {us[3]:5.2f} total {us[4]:6.2f}", flush=True)


NI = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
if len(sys.argv) > 2:
    ops.set_stream(sys.argv[2] != '0')
for (H, Cin, Cout) in [(96, 64, 256), (96, 256, 64), (48, 128, 512), (48, 512, 128), (24, 256, 1024), (24, 1024, 256), (12, 512, 2048)]:
    W = H
    rows = NI * H * W
    x = torch.randn(NI, H, W, Cin, device=dev); w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    img = ops.build_weight_planes(w, Cout, 1, Cin)
    byt = 4.0 * rows * (Cin + Cout)
    for st in (True, False):
        t, b = timeit(lambda: ops.conv2d_fwd(x, w, NI, H, W, Cin, Cout, 1, 1, 1, 0, sc, sh, stats=st, wimg=img))
        show(f"fwd {Cin}->{Cout} @{H} stats={int(st)}", t, b, byt)
    t, b = timeit(lambda: ops.conv2d_fwd(x, w, NI, H, W, Cin, Cout, 1, 1, 1, 0, None, None, stats=True, wimg=img))
    show(f"fwd {Cin}->{Cout} @{H} no prologue, stats=1", t, b, byt)
    # data gradient of the same layer: dy [rows, Cout] formed from (dz, c) in the loader; epilogue reduces the BatchNorm behind x
    g = torch.randn(NI, H, W, Cout, device=dev) * 1e-3; c = torch.randn(NI, H, W, Cout, device=dev)
    gam, bet = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    rm, rv, nbt = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    saved = ops.bn_finalize(ops.colstats(c, rows, Cout), Cout, rows, gam, bet, rm, rv, nbt, 0.1, 1e-5, True)
    dg, db = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    ap = ops.bn_bwd(g, c, saved, rows, Cout, rows, dg, db, 2, fused=True)
    cx = torch.randn(NI, H, W, Cin, device=dev)
    gx, bx = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    rmx, rvx = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    savedx = ops.bn_finalize(ops.colstats(cx, rows, Cin), Cin, rows, gx, bx, rmx, rvx, nbt, 0.1, 1e-5, True)
    res = torch.randn(NI, H, W, Cin, device=dev)
    byd = 4.0 * rows * (2 * Cout + Cin)
    t, b = timeit(lambda: ops.conv2d_dgrad(ap, w, NI, H, W, Cin, Cout, 1, 1, 1, 0, wimg=img))
    show(f"dgrad {Cout}->{Cin} @{H} apply", t, b, byd)
    t, b = timeit(lambda: ops.conv2d_dgrad(ap, w, NI, H, W, Cin, Cout, 1, 1, 1, 0, wimg=img,
                                           bnb=dict(mode=2, c=cx, saved=savedx, dz_amax=True)))
    show(f"dgrad {Cout}->{Cin} @{H} apply +bnb(mode 2)", t, b, byd + 4.0 * rows * Cin)
    t, b = timeit(lambda: ops.conv2d_dgrad(ap, w, NI, H, W, Cin, Cout, 1, 1, 1, 0, residual=res, wimg=img,
                                           bnb=dict(mode=1, c=cx, y=res, saved=savedx, dz_amax=True)))
    show(f"dgrad {Cout}->{Cin} @{H} apply +res +bnb(mode 1)", t, b, byd + 4.0 * rows * Cin * 3)
    del x, g, c, cx, res, ap
