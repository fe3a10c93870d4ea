"""microbench: the element-wise / reduction kernels of the trunk at layer-1 size (320 slices of 96x96, 256 channels)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rows, C in [(320 * 96 * 96, 256), (320 * 96 * 96, 64), (320 * 24 * 24, 1024)]:
    c = torch.randn(rows, C, device=dev); idt = torch.randn(rows, C, device=dev); g = torch.randn(rows, C, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    saved = ops.bn_finalize(ops.colstats(c, rows, C), C, rows, gam, bet, rm, rv, nbt, 0.1, 1e-5, True)
    out = torch.empty_like(c)
    GB = rows * C * 4 / 1e9
    t = timeit(lambda: ops.bn_add_relu(c, saved, rows, C, idt=idt, out=out)); print(f"rows {rows} C {C}: bn_add_relu      {t:7.3f} ms  {3*GB/t:6.2f} TB/s (3 tensors)")
    t = timeit(lambda: ops.colstats(c, rows, C)); print(f"                       colstats         {t:7.3f} ms  {GB/t:6.2f} TB/s (1 tensor)")
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    gg = g.clone()
    t = timeit(lambda: ops.bn_bwd(gg, c, saved, rows, C, rows, dg, db, 2, fused=True)); print(f"                       bn_bwd (fused)   {t:7.3f} ms  {3*GB/t:6.2f} TB/s (read g, c; write dz)")
    t = timeit(lambda: ops.act_planes(c, rows, C, 1, saved[2], saved[3], fscale=16.0)); print(f"                       act_planes tf1   {t:7.3f} ms  {2*GB/t:6.2f} TB/s (read 1, write 1)")
    t = timeit(lambda: out.copy_(c)); print(f"                       torch copy       {t:7.3f} ms  {2*GB/t:6.2f} TB/s")
