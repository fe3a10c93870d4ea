"""GEMM launches for `rocprofv3 --pmc` counter collection (scripts/collect_sq_counters.sh -> profiles/r02_gemm_sq_counters.json):
REP launches each, after a warm-up, on random operands (zeros clock higher), of
  * dense forward 4096^3 on the bf16 x 3 scheme (linear layers; the round-1 reference point),
  * the 3x3 256->256 convolution of the synthetic-shape slices (160 slices of 24 x 24: 92 160 pixels) on the fp16 x 2 scheme --
    with the fp32 loaders (forward: BatchNorm prologue + statistics epilogue, weight tiles DMA'd from the plane image; data
    gradient: weight image, dy formed from (dz, c) in the loader; weight gradient: same dy), with the activation operands
    from plane images (forward through the per-tap gather kernel and through the halo kernel, weight gradient through the
    K-major pair) -- and the same forward on the bf16 x 3 scheme."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M, N, K = 4096, 4096, 4096
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
for _ in range(REP): ops.linear_fwd(x, w, None, M, N, K)
N_, H, W, Cin, Cout, k, s, p = 160, 24, 24, 256, 256, 3, 1, 1
rows = N_ * H * W
xx = torch.randn(N_, H, W, Cin, device=dev); ww = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
g = torch.randn(N_, H, W, Cout, device=dev) * 1e-3; c = torch.randn(N_, H, W, Cout, device=dev); dw = torch.empty_like(ww)
sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
img = ops.build_weight_planes(ww, Cout, k * k, Cin)
gam, bet = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
rm, rv, nbt = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
saved = ops.bn_finalize(ops.colstats(c, rows, Cout), Cout, rows, gam, bet, rm, rv, nbt, 0.1, 1e-5, True)
dg, db = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
ap = ops.bn_bwd(g, c, saved, rows, Cout, rows, dg, db, 2, fused=True)
# fp32 loaders (round-2 first half): activation operand converted in the k-loop
for _ in range(REP): ops.conv2d_fwd(xx, ww, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=img, aplanes=False)
for _ in range(REP): ops.conv2d_dgrad(ap, ww, N_, H, W, Cin, Cout, k, k, s, p, wimg=img, aplanes=False)
for _ in range(REP): ops.conv2d_wgrad(ap, xx, dw, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, aplanes=False)
# activation plane images: per-tap gather kernel, halo kernel (forward), K-major pair (weight gradient)
ops.set_conv3x3_halo(0)
for _ in range(REP): ops.conv2d_fwd(xx, ww, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=img, aplanes=True)
ops.set_conv3x3_halo(1)
for _ in range(REP): ops.conv2d_fwd(xx, ww, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=img, aplanes=True)
for _ in range(REP): ops.conv2d_wgrad(ap, xx, dw, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, aplanes=True)
# the same forward on the bf16 x 3 scheme
for _ in range(REP): ops.conv2d_fwd(xx, ww, N_, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True)
torch.cuda.synchronize()
