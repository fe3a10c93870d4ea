"""(GPU box) the streamed 1x1 kernels (KoafGemm A mode M_KS, koaf_set_stream) against the block-wide loader: bit-identity of every
output and the time of both, on the 1x1 layers of the synthetic-shape ResNet-50 trunk
    python scripts/dev_stream.py [slices=320]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops

dev = torch.device("cuda:0")
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 320
torch.manual_seed(0)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def both(fn):
    """fn() -> tuple of tensors; run with the block-wide loader and streamed; -> (identical, ms old, ms new)"""
    was = ops.set_stream(False)
    try:
        r0 = [t.clone() for t in fn() if torch.is_tensor(t)]
        t0 = timeit(fn)
        ops.set_stream(True)
        r1 = [t.clone() for t in fn() if torch.is_tensor(t)]
        t1 = timeit(fn)
    finally:
        ops.set_stream(was)
    same = len(r0) == len(r1) and all(torch.equal(a, b) for a, b in zip(r0, r1))
    if not same:
        for i, (a, b) in enumerate(zip(r0, r1)):
            if not torch.equal(a, b):
                d = (a.double() - b.double()).abs().max().item() / max(a.double().abs().max().item(), 1e-30)
                print(f"      output {i} {tuple(a.shape)} differs: max |d| / max |a| = {d:.2e}", flush=True)
    return same, t0, t1


def bn(C, rows, x):
    return ops.bn_finalize(ops.colstats(x, rows, C), C, rows, (torch.randn(C, device=dev) * 0.2 + 1), torch.randn(C, device=dev) * 0.1,
                           torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), 0.1, 1e-5, True)


ok = True
# forward: (H, Cin, Cout, kind)   kind: 0 plain, 1 BatchNorm prologue, 3 tail, 4 tail behind a downsample branch
fwd = [(96, 64, 256, 1), (48, 128, 512, 1), (24, 256, 1024, 1), (12, 512, 2048, 1), (96, 64, 64, 0), (96, 256, 64, 3), (48, 512, 128, 3),
       (24, 1024, 256, 3), (12, 2048, 512, 3), (96, 256, 128, 4), (24, 1024, 512, 4), (96, 256, 64, 5)]
for (H, Cin, Cout, kind) in fwd:
    N_ = NS
    rows = N_ * H * H
    x = torch.randn(N_, H, H, Cin, device=dev)
    w = torch.randn(Cout, 1, 1, Cin, device=dev) * Cin ** -0.5
    img = ops.build_weight_planes(w, Cout, 1, Cin)
    sv = bn(Cin, rows, x)
    shift = torch.randn(Cout, device=dev) * 0.1
    if kind in (0, 1):
        sc, sh = (sv[2], sv[3]) if kind else (None, None)
        fn = lambda: ops.conv2d_fwd(x, w, N_, H, H, Cin, Cout, 1, 1, 1, 0, sc, sh, stats=True, shift=shift, wimg=img)
        tag = "bn-prologue" if kind else "plain"
    elif kind == 5:
        # (rebuilt stage: known BatchNorm behind the convolution, the epilogue cuts the consumer's plane images)
        idt = torch.randn(N_, H, H, Cin, device=dev)
        em = (torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1)

        def fn():
            y, part, yin = ops.conv2d_fwd(x, w, N_, H, H, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=False, wimg=img, tail_idt=idt, emit=em)
            return y, yin, y._koaf_eplanes[0]
        tag = "tail + emit"
    else:
        idt = torch.randn(N_, H, H, Cin, device=dev)
        ids = bn(Cin, rows, idt) if kind == 4 else None
        fn = lambda: ops.conv2d_fwd(x, w, N_, H, H, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=True, shift=shift, wimg=img, tail_idt=idt,
                                    tail_idsaved=ids)
        tag = "tail" + (" (downsample identity)" if kind == 4 else "")
    same, t0, t1 = both(fn)
    ok &= same
    by = 4.0 * rows * (Cin * (2 if kind >= 3 else 1) + Cout + (Cin if kind >= 3 else 0))
    print(f"fwd 1x1 {Cin:5d}->{Cout:5d} px{rows:9d} {tag:28s} identical {same}  block-wide {t0:7.3f} ms  streamed {t1:7.3f} ms  ({t0/t1:4.2f}x, {by/t1/1e9:5.2f} TB/s)", flush=True)
    del x

# data gradient with the BatchNorm-backward apply on load and the fused reduction of the producer's BatchNorm (+ residual)
dg = [(96, 64, 256), (96, 256, 64), (48, 128, 512), (48, 512, 128), (24, 256, 1024), (24, 1024, 256), (12, 512, 2048), (12, 2048, 512)]
for (H, Cin, Cout) in dg:          # convolution Cin -> Cout: the data gradient contracts over Cout
    N_ = NS
    rows = N_ * H * H
    c = torch.randn(N_, H, H, Cout, device=dev) * 1.5 + 0.3
    g = torch.randn(N_, H, H, Cout, device=dev) * 1e-3
    w = torch.randn(Cout, 1, 1, Cin, device=dev) * Cin ** -0.5
    img = ops.build_weight_planes(w, Cout, 1, Cin)
    saved = bn(Cout, rows, c)
    dgm, dbt = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    ap = ops.bn_bwd(g.clone(), c, saved, rows, Cout, rows, dgm, dbt, 2, fused=True)
    cx = torch.randn(N_, H, H, Cin, device=dev) * 1.5 + 0.3
    savx = bn(Cin, rows, cx)
    res = torch.randn(N_, H, H, Cin, device=dev) * 1e-3
    for bnb, tag in ((None, "apply"), (dict(mode=2, c=cx, saved=savx, dz_amax=True), "apply + bnb + residual")):
        fn = lambda: ops.conv2d_dgrad(ap, w, N_, H, H, Cin, Cout, 1, 1, 1, 0, residual=res if bnb else None, bnb=bnb, wimg=img)
        fn2 = (lambda: fn()) if bnb else (lambda: (fn(),))
        same, t0, t1 = both(fn2)
        ok &= same
        by = 4.0 * rows * (2 * Cout + Cin * (4 if bnb else 1))
        print(f"dgrad 1x1 {Cout:5d}->{Cin:5d} px{rows:9d} {tag:26s} identical {same}  block-wide {t0:7.3f} ms  streamed {t1:7.3f} ms  ({t0/t1:4.2f}x, {by/t1/1e9:5.2f} TB/s)", flush=True)
    del c, g, cx, res
print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
