"""Throughput of the device input pipeline (preproc.PTBatchAugment: per-sample min/max + one fused kernel) on the native
shapes, against the HBM roofline (algorithmic bytes: the raw batch read twice -- min/max pass and augment pass -- and the
result written once).  GPU box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from oaprogressionmmf_amd.preproc import PTBatchAugment

dev = torch.device("cuda:0")
HBM_PEAK_GBS = 8000.0


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for tag, shape in (("xr 700x700", (8, 1, 700, 700)), ("dess 320x320x128", (8, 1, 320, 320, 128)), ("t2 320x320x25", (8, 1, 320, 320, 25))):
    x = torch.rand(shape, device=dev) * 1000 + 3
    for name, aug in (("train (rotate+gamma drawn)", PTBatchAugment(mean=0.3, std=0.25)),
                      ("eval (unit range + normalise)", PTBatchAugment(mean=0.3, std=0.25, rotate_prob=0.0, gamma_prob=0.0))):
        states = [(0.1, 0.2, 0.1, 1.4)] * shape[0] if "train" in name else None
        ms = timeit(lambda: aug(x, states=states) if states else aug(x))
        gb = 3 * x.numel() * 4 / 1e9
        print(f"{tag:18s} {name:30s} {ms:7.3f} ms  {shape[0] / ms * 1e3:9.0f} samples/s  {gb / ms * 1e3:7.0f} GB/s "
              f"= {gb / ms * 1e3 / HBM_PEAK_GBS:.2f} of HBM peak")
