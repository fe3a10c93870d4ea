"""Forward attention of the FeaT fusion at the headline shape, one launch (n <= 512) vs the three-launch path
(KOAF_ATTN_FUSED=0 python scripts/bench_attention.py for the latter; the switch is read once per process)."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
for (B, n, h, d) in [(8, 483, 8, 256), (8, 161, 8, 256), (8, 92, 8, 256), (8, 25, 8, 256)]:
    qkv = torch.randn(B, n, 3 * h * d, device=dev)
    scale = (h * d) ** -0.5
    for _ in range(3): ops.attention_fwd(qkv, B, n, h, d, scale)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.attention_fwd(qkv, B, n, h, d, scale)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    fl = 4.0 * B * h * n * n * d
    print(f"fused={os.environ.get('KOAF_ATTN_FUSED', '1')} B={B} n={n} h={h} d={d}: {t * 1e3:8.1f} us  {fl / t / 1e9:6.1f} TF/s", flush=True)
