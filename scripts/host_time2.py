"""host enqueue time vs GPU time of the default bench step (GPU box)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch, procedural as P
from oaprogressionmmf_amd.config import ConfigDict
from oaprogressionmmf_amd.models import dict_models
from oaprogressionmmf_amd.various import dict_losses, dict_optimizers
dev = torch.device("cuda:0")
cfg = P.cfg_xr1mr3c1(dropout=0.1); B = 8
m = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None).to(dev).train()
loss_fn = dict_losses["FocalLoss"](reduction="mean", gamma=2.0, num_classes=2)
opt = dict_optimizers["Adam"](m.parameters(), lr=1e-4, weight_decay=1e-4)
xs = [torch.from_numpy(a).to(dev) for a in P.model_inputs(cfg, B)]
y = torch.from_numpy(P.make_target("target", B)).to(dev)
def step():
    opt.zero_grad()
    t0 = time.perf_counter()
    lg = m(*xs)["main"]
    t1 = time.perf_counter()
    loss = loss_fn(input=lg.squeeze(1), target=y.long().squeeze(1))
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(3): step()
torch.cuda.synchronize()
acc = [0, 0, 0]; n = 8
T0 = time.perf_counter()
for _ in range(n):
    a = step()
    torch.cuda.synchronize()
    acc = [x + y_ for x, y_ in zip(acc, a)]
T = (time.perf_counter() - T0) / n
print(f"step {T*1e3:.1f} ms; host enqueue: fwd {acc[0]/n*1e3:.1f} bwd {acc[1]/n*1e3:.1f} opt {acc[2]/n*1e3:.1f} ms")
