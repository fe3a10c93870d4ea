"""GEMM launches of the SHORT-K 3x3 layers for `rocprofv3 --pmc` counter collection (scripts/collect_sq_counters.sh short ->
profiles/r03_gemm_sq_counters_short.json): the 64 -> 64 convolution of layer1 on 96 x 96 images and the 128 -> 128 one of
layer2 on 48 x 48 (320 slices each, random operands), every kernel the train step runs on them --
  forward: halo kernel, 128-row / two blocks per CU shape (what the planner picks for 64 channels) and 256-row pipelined shape;
  data gradient (dy from plane images with the BatchNorm-backward apply, fused BatchNorm-backward reduction in the epilogue);
  weight gradient (both operands from plane images, K-major);
and, for the 1x1 family around them, conv3 64 -> 256 forward (statistics epilogue) and conv1 256 -> 64 with the bottleneck tail
formed on load.  The launch plan (label, kernel template, launches) is written beside the counters: the summary splits each
template's dispatches by it."""
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


for (N_, H, W, C) in ((320, 96, 96, 64), (320, 48, 48, 128)):
    rows = N_ * H * W
    bn = 64 if C == 64 else 128
    x = torch.randn(N_, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    img = ops.build_weight_planes(w, C, 9, C)
    xpl = ops.act_planes(x, rows, C, 1, sc, sh, fscale=16.0)
    tag = f"3x3 {C}->{C} @{H}"
    for mode, shape in ((3, "128, %d, 9, 6, 0, 0, true, true, 256, 0" % bn), (2, "256, %d, 9, 6, 0, 0, true, true, 512, 0" % bn)):
        ops.set_conv3x3_halo(mode)
        run(f"{tag} forward, halo kernel {shape.split(',')[0]}-row shape", shape,
            lambda: ops.conv2d_fwd(x, w, N_, H, W, C, C, 3, 3, 1, 1, sc, sh, stats=True, wimg=img, aplanes=xpl))
    ops.set_conv3x3_halo(1)
    g = torch.randn(N_, H, W, C, device=dev) * 1e-3; c = torch.randn(N_, H, W, C, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    saved = ops.bn_finalize(ops.colstats(c, rows, C), C, rows, gam, bet, rm, rv, nbt, 0.1, 1e-5, True)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ap = ops.bn_bwd(g, c, saved, rows, C, rows, dg, db, 2, fused=True)
    cx = torch.randn(N_, H, W, C, device=dev)
    savedx = ops.bn_finalize(ops.colstats(cx, rows, C), C, rows, gam, bet, rm.clone(), rv.clone(), nbt, 0.1, 1e-5, True)
    dypl = ops._dy_planes(ap, None, rows, C)
    dshape = ("128, %d, 9, 6, 0, 0, true, true, 256, 2" % bn) if C == 64 else ("256, %d, 9, 6, 0, 0, true, true, 512, 2" % bn)
    # (act16 = 0 here: the fp32 mode's data gradient is the <..., 0> instantiation)
    dshape = dshape[:-1] + "0"
    run(f"{tag} data gradient (+ fused BatchNorm-backward reduction), halo kernel", dshape,
        lambda: ops.conv2d_dgrad(ap, w, N_, H, W, C, C, 3, 3, 1, 1, wimg=img, bnb=dict(mode=2, c=cx, saved=savedx, dz_amax=True)))
    dw = torch.empty_like(w)
    wshape = ("64, 128, 10, 11, 0, 0, true, true, 256, 0" if C == 64 else "128, 128, 10, 11, 0, 0, true, true, 256, 0")
    run(f"{tag} weight gradient, both operands from plane images (K-major)", wshape,
        lambda: ops.conv2d_wgrad(ap, x, dw, N_, H, W, C, C, 3, 3, 1, 1, sc, sh))
# the 1x1 convolutions around layer1's 3x3
N_, H, W = 320, 96, 96
rows = N_ * H * W
x64 = torch.randn(N_, H, W, 64, device=dev); w3 = torch.randn(256, 1, 1, 64, device=dev) * 0.1
img3 = ops.build_weight_planes(w3, 256, 1, 64)
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
run("1x1 64->256 @96 forward (conv3: BatchNorm prologue on load, statistics epilogue), persistent fp32 loader",
    "128, 128, 0, 6, 1, 0, true, true, 256, 0", lambda: ops.conv2d_fwd(x64, w3, N_, H, W, 64, 256, 1, 1, 1, 0, sc, sh, stats=True, wimg=img3))
c3 = torch.randn(N_, H, W, 256, device=dev); idt = torch.relu(torch.randn(N_, H, W, 256, device=dev))
w1 = torch.randn(64, 1, 1, 256, device=dev) * 0.05
img1 = ops.build_weight_planes(w1, 64, 1, 256)
s3 = torch.stack([torch.zeros(256), torch.ones(256), torch.ones(256), torch.zeros(256)]).to(dev)
yb = torch.empty_like(c3)
run("1x1 256->64 @96 forward (conv1) with the bottleneck tail relu(bn3(c3) + identity) formed on load + side store of y",
    "128, 64, 0, 6, 3, 0, true, true, 256, 0",
    lambda: ops.conv2d_fwd(c3, w1, N_, H, W, 256, 64, 1, 1, 1, 0, s3[2], s3[3], stats=True, wimg=img1, tail_idt=idt, tail_out=yb))
torch.cuda.synchronize()
out = Path(sys.argv[2]) if len(sys.argv) > 2 else Path("/tmp/pmc_short_plan.json")
out.write_text(json.dumps(PLAN, indent=1))
