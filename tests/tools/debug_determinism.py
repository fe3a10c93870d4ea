"""two train-mode forward(+backward) passes of the native3 model from the same state: losses / gradients must be bit-identical"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import procedural as P
from oaprogressionmmf_amd.config import ConfigDict
from oaprogressionmmf_amd.models import dict_models
from oaprogressionmmf_amd.various import dict_losses

dev = torch.device("cuda:0")
cfg = P.cfg_xr1mr3c1(dropout=0.0)
torch.manual_seed(5)
m = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bwd = len(sys.argv) > 2
xs = [torch.from_numpy(a).to(dev) for a in P.model_inputs(cfg, B, 1234)]
y = torch.from_numpy(P.make_target("target", B, 1234)).to(dev)
loss_fn = dict_losses["FocalLoss"](reduction="mean", gamma=2.0, num_classes=2)
buf0 = {k: b.detach().clone() for k, b in m.named_buffers()}
res = []
for it in range(4):
    with torch.no_grad():
        for k, b in m.named_buffers():
            b.copy_(buf0[k])
    m.train(); m.zero_grad()
    out = m(*xs)["main"]
    loss = loss_fn(input=out.squeeze(1), target=y.long().squeeze(1))
    if bwd:
        loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None} if bwd else {}
    res.append((float(loss), out.detach().clone(), g))
    print(it, float(loss), flush=True)
for it in range(1, 4):
    same = torch.equal(res[0][1], res[it][1])
    bad = [k for k in res[0][2] if not torch.equal(res[0][2][k], res[it][2][k])]
    print("run", it, "logits identical:", same, "grad tensors differing:", len(bad), bad[:5])
