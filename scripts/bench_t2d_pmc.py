"""GEMM launches of layer1's 3x3 64 -> 64 convolution (320 slices of 96 x 96) for `rocprofv3 --pmc` counter collection
(scripts/collect_sq_counters.sh <dir> t2d -> profiles/r04_gemm_sq_counters_short.json): the round-4 rectangle-tile kernel
(KoafGemm mode M_PT) beside the 128-row raster halo kernel it replaces, forward and data gradient (dy from plane images with the
BatchNorm-backward apply, fused BatchNorm-backward reduction in the epilogue).  Writes its launch plan beside the counters."""
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 12
PLAN = []


def run(label, template, fn):
    for _ in range(REP):
        fn()
    PLAN.append({"label": label, "template": template, "launches": REP})


N_, H, W, C = 320, 96, 96, 64
rows = N_ * H * W
x = torch.randn(N_, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
img = ops.build_weight_planes(w, C, 9, C)
xpl = ops.act_planes(x, rows, C, 1, sc, sh, fscale=16.0)
g = torch.randn(N_, H, W, C, device=dev) * 1e-3; c = torch.randn(N_, H, W, C, device=dev)
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
saved = ops.bn_finalize(ops.colstats(c, rows, C), C, rows, gam, bet, rm, rv, nbt, 0.1, 1e-5, True)
dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
ap = ops.bn_bwd(g, c, saved, rows, C, rows, dg, db, 2, fused=True)
cx = torch.randn(N_, H, W, C, device=dev)
savedx = ops.bn_finalize(ops.colstats(cx, rows, C), C, rows, gam, bet, rm.clone(), rv.clone(), nbt, 0.1, 1e-5, True)
for mode, shape, name in ((3, "128, 64, 9, 6, 0, 0, true, true, 256, 0", "128-row raster halo kernel (round 3)"),
                          (1, "128, 64, 12, 6, 0, 0, true, true, 256, 0", "8 x 16 rectangle-tile kernel (round 4, M_PT)")):
    ops.set_conv3x3_halo(mode)
    run(f"3x3 64->64 @96 forward, {name}", shape,
        lambda: ops.conv2d_fwd(x, w, N_, H, W, C, C, 3, 3, 1, 1, sc, sh, stats=True, wimg=img, aplanes=xpl))
    run(f"3x3 64->64 @96 data gradient (+ fused BatchNorm-backward reduction), {name}", shape,
        lambda: ops.conv2d_dgrad(ap, w, N_, H, W, C, C, 3, 3, 1, 1, wimg=img, bnb=dict(mode=2, c=cx, saved=savedx, dz_amax=True)))
ops.set_conv3x3_halo(1)
torch.cuda.synchronize()
if len(sys.argv) > 2:
    json.dump(PLAN, open(sys.argv[2], "w"), indent=1)
print("done", len(PLAN))
