"""development check of the 2-D tile 3x3 kernel (KoafGemm M_PT): forward and data gradient against float64 and against the
128-row halo kernel; timing on the synthetic-shape layer1 / layer2 convolutions.  gpurun -- python scripts/dev_t2d.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.nn.functional as F
from oaprogressionmmf_amd import ops

dev = torch.device("cuda:0")
G = torch.Generator().manual_seed(0)


def rnd(*shape, scale=1.0):
    return torch.randn(*shape, generator=G) * scale


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(N, H, W, Cin, Cout):
    x = rnd(N, H, W, Cin).to(dev)
    w = rnd(Cout, 3, 3, Cin, scale=(9 * Cin) ** -0.5).to(dev)
    sc, sh = (rnd(Cin) * 0.2 + 1).to(dev), (rnd(Cin) * 0.1).to(dev)
    img = ops.build_weight_planes(w, Cout, 9, Cin)
    xin = torch.relu(x.double().cpu() * sc.double().cpu() + sh.double().cpu())
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    out = {}
    for mode in (3, 1):
        was = ops.set_conv3x3_halo(mode)
        try:
            y, part = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, 3, 3, 1, 1, sc, sh, stats=True, wimg=img, aplanes=True)
        finally:
            ops.set_conv3x3_halo(was)
        out[mode] = (y, part)
        print(f"fwd {N}x{H}x{W} {Cin}->{Cout} mode {mode}: vs f64 {rel(y.cpu(), ref):.2e}  stats rows {part.shape[0]}")
    print("   t2d vs halo128:", rel(out[1][0].cpu(), out[3][0].cpu()), " stats totals:", rel(out[1][1].sum(0).cpu(), out[3][1].sum(0).cpu()))
    # data gradient (flipped taps, D image), plain tensor dy with its amax
    dy = rnd(N, H, W, Cout).to(dev)
    am = dy.abs().max().reshape(1).float()
    refdx = F.conv_transpose2d(dy.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    for mode in (3, 1):
        was = ops.set_conv3x3_halo(mode)
        try:
            dx = ops.conv2d_dgrad(dy, w, N, H, W, Cin, Cout, 3, 3, 1, 1, wimg=img, dy_amax=am)
        finally:
            ops.set_conv3x3_halo(was)
        print(f"dgrad mode {mode}: vs f64 {rel(dx.cpu(), refdx):.2e}")


def bench(N, H, W, C, reps=5):
    x = torch.randn(N, H, W, C, device=dev)
    w = (torch.randn(C, 3, 3, C, device=dev) * (9 * C) ** -0.5)
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    img = ops.build_weight_planes(w, C, 9, C)
    xpl = ops.act_planes(x, N * H * W, C, 1, sc, sh, fscale=ops.ACT_SCALE)
    for mode in (3, 1):
        was = ops.set_conv3x3_halo(mode)
        try:
            for _ in range(2):
                ops.conv2d_fwd(x, w, N, H, W, C, C, 3, 3, 1, 1, sc, sh, stats=True, wimg=img, aplanes=xpl)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv2d_fwd(x, w, N, H, W, C, C, 3, 3, 1, 1, sc, sh, stats=True, wimg=img, aplanes=xpl)
            e1.record()
            torch.cuda.synchronize()
        finally:
            ops.set_conv3x3_halo(was)
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * N * H * W * C * 9 * C
        print(f"bench fwd {N}x{H}x{W}x{C} mode {mode}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    for shp in ((1, 8, 16, 64, 64), (3, 96, 96, 64, 64), (2, 48, 48, 128, 128), (3, 24, 32, 128, 64), (2, 16, 16, 64, 128)):
        check(*shp)
    bench(1280, 96, 96, 64)
    bench(1280, 48, 48, 128)
