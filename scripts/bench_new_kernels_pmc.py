"""Launches of the matrix-pipe kernels added in the second half of round 3, for `rocprofv3 --pmc` counter collection
(scripts/collect_sq_counters.sh <dir> new -> profiles/r03_sq_counters_new_kernels.json): the 3x3 weight gradient in padded raster
order (wgrad3x3_ring_kernel: 64 -> 64 @ 96^2, 128 -> 128 @ 48^2, 256 -> 256 @ 24^2, 320 slices, plane images cut beforehand),
the stem forward (with the statistics epilogue) and weight gradient (with the BatchNorm-backward apply on load) on 320 slices of
384^2, and the single-launch attention forward at the fusion's shape.  The launch plan (label, kernel name, launches) is written
beside the counters."""
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 12
PLAN = []


def run(label, kernel, fn):
    for _ in range(REP):
        fn()
    PLAN.append({"label": label, "kernel": kernel, "launches": REP})


for (N_, H, W, C) in ((320, 96, 96, 64), (320, 48, 48, 128), (320, 24, 24, 256)):
    rows = N_ * H * W
    x = torch.randn(N_, H, W, C, device=dev); dy = torch.randn(N_, H, W, C, device=dev) * 1e-3
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    am = dy.abs().max().reshape(1)
    L = ops.lib()
    xpl = ops.act_planes(x, rows, C, 1, sc, sh, fscale=ops.ACT_SCALE)
    dypl = ops.act_planes(dy, rows, C, 0, amax=am)
    dw = torch.empty(C, 3, 3, C, device=dev)
    slabs = torch.empty(L.koaf_conv2d_wgrad_ws(N_, H, W, C, C, 3, 3, 1, 1), device=dev)
    run(f"3x3 {C}->{C} @{H} weight gradient, padded raster order (x in an LDS ring)", "wgrad3x3_ring_kernel",
        lambda: ops.check(L.koaf_conv2d_wgrad(None, ops._ptr(x), ops._ptr(dw), N_, H, W, C, C, 3, 3, 1, 1, None, None, ops._ptr(slabs),
                                              ops._ptr(am), None, dypl.data_ptr(), xpl.data_ptr(), 0, ops._stream()), "wgrad"))
    del x, dy, xpl, dypl, slabs
N_, H, W = 320, 384, 384
x = torch.randn(N_, H, W, device=dev); w1t = torch.randn(49, 64, device=dev) * 0.1
shift = torch.zeros(64, device=dev)
run("stem 7x7/s2 forward + BatchNorm statistics, matrix pipe (3 bf16 pieces)", "stem_fwd_mma_kernel",
    lambda: ops.stem_fwd(x, w1t, N_, H, W, stats=True, shift=shift))
y = ops.stem_fwd(x, w1t, N_, H, W)
H1, W1, C = y.shape[1], y.shape[2], 64
rows = N_ * H1 * W1
saved = ops.bn_finalize(ops.colstats(y, rows, C), C, rows, torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev),
                        torch.ones(C, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), 0.1, 1e-5, True)
yp, amx = ops.maxpool_fwd(y, saved, N_, H1, W1, C)
dyp = torch.randn_like(yp)
dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
ap = ops.bn_bwd(None, y, saved, rows, C, rows, dg, db, 2, fused=True, pool=(dyp, amx, N_, H1, W1))
dw = torch.empty(64, 7, 7, 3, device=dev)
run("stem weight gradient, matrix pipe, dc formed on load (BatchNorm-backward apply)", "stem_wgrad_mma_kernel",
    lambda: ops.stem_wgrad(ap, x, dw, N_, H, W))
B, n, h, d = 8, 483, 8, 256
qkv = torch.randn(B, n, 3 * h * d, device=dev)
run("attention forward, one launch (n = 483, 8 heads x 256)", "attention_fwd_kernel", lambda: ops.attention_fwd(qkv, B, n, h, d, (h * d) ** -0.5))
torch.cuda.synchronize()
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(json.dumps(PLAN, indent=1))
