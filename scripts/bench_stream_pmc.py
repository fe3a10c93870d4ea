"""GEMM launches of the 1x1 convolutions of the synthetic-shape trunk (320 slices) for `rocprofv3 --pmc` counter collection
(scripts/collect_sq_counters.sh <dir> stream -> profiles/r04_sq_counters_stream.json): the streamed kernel (KoafGemm A mode M_KS,
round 4) beside the block-wide loader it replaces, same calls -- forward with the BatchNorm prologue, forward with the bottleneck
tail, data gradient with the BatchNorm-backward apply and the fused reduction.  Writes its launch plan beside the counters."""
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 10
PLAN = []
NS = 320


def run(label, template, fn):
    for _ in range(REP):
        fn()
    PLAN.append({"label": label, "template": template, "launches": REP})


def bn(C, rows, x):
    return ops.bn_finalize(ops.colstats(x, rows, C), C, rows, torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev),
                           torch.ones(C, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), 0.1, 1e-5, True)


for (H, Cin, Cout) in [(48, 128, 512), (24, 256, 1024), (24, 1024, 256)]:
    rows = NS * H * H
    x = torch.randn(NS, H, H, Cin, device=dev); w = torch.randn(Cout, 1, 1, Cin, device=dev) * Cin ** -0.5
    img = ops.build_weight_planes(w, Cout, 1, Cin)
    sv = bn(Cin, rows, x)
    idt = torch.randn(NS, H, H, Cin, device=dev)
    g = torch.randn(NS, H, H, Cin, device=dev) * 1e-3
    wd = torch.randn(Cin, 1, 1, Cout, device=dev) * Cout ** -0.5
    imgd = ops.build_weight_planes(wd, Cin, 1, Cout)
    dgm, dbt = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev)
    ap = ops.bn_bwd(g.clone(), x, sv, rows, Cin, rows, dgm, dbt, 2, fused=True)
    cx = torch.randn(NS, H, H, Cout, device=dev)
    savx = bn(Cout, rows, cx)
    for stream, am, sd, name in ((False, 0, 0, "block-wide loader"), (True, 13, 2, "streamed (M_KS)")):
        was = ops.set_stream(stream)
        run(f"1x1 {Cin}->{Cout} @{H} forward, BatchNorm prologue + statistics, {name}", f"128, 128, {am}, 6, 1, 0, true, true, 256, 0, false, {sd}",
            lambda: ops.conv2d_fwd(x, w, NS, H, H, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=True, wimg=img))
        run(f"1x1 {Cin}->{Cout} @{H} forward, bottleneck tail on load, {name}", f"128, 128, {am}, 6, 3, 0, true, true, 256, 0, false, {sd}",
            lambda: ops.conv2d_fwd(x, w, NS, H, H, Cin, Cout, 1, 1, 1, 0, sv[2], sv[3], stats=True, wimg=img, tail_idt=idt))
        run(f"1x1 data gradient {Cin}->{Cout} @{H} (apply on load + fused BatchNorm-backward reduction), {name}",
            f"128, 128, {am}, 6, 2, 0, true, true, 256, 0, false, {sd}",
            lambda: ops.conv2d_dgrad(ap, wd, NS, H, H, Cout, Cin, 1, 1, 1, 0, wimg=imgd, bnb=dict(mode=2, c=cx, saved=savx, dz_amax=True)))
        ops.set_stream(was)
    del x, idt, g, cx
torch.cuda.synchronize()
if len(sys.argv) > 2:
    json.dump(PLAN, open(sys.argv[2], "w"), indent=1)
print("done", len(PLAN))
