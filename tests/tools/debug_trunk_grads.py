"""debug: per-parameter gradient error of a KoafTrunk vs the oracle, last layers first"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import procedural as P
from oracle import koafusion_cpu as O
from oaprogressionmmf_amd.models._core_fes import dict_fes
from oaprogressionmmf_amd.models._encoder import KoafTrunk

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
shape = (2, 1, 96, 112)
dev = torch.device("cuda:0")
net = dict_fes[arch](pretrained=False)
trunk = KoafTrunk(*list(net.children())[:-1])
P.fill_state_dict(trunk.state_dict())
trunk = trunk.to(dev).train()
x = torch.from_numpy(P.make_input("trunk", shape))
gy = None
y = trunk(x.to(dev))
gy = torch.from_numpy(P.make_input("trunkg", tuple(y.shape)))
(y * gy.to(dev)).sum().backward()
spec = O.trunk_spec("t", arch)
sd = {k: torch.from_numpy(P.fill_value(k[2:], s, dt == torch.int64)).reshape(s) for k, s, dt in spec}
for k in sd:
    if O.is_param(k): sd[k].requires_grad_(True)
yo = O.trunk(x, sd, "t", arch, True)
(yo * gy).sum().backward()
print("fwd rel", ((y.detach().cpu() - yo.detach()).norm() / yo.detach().norm()).item())
rows = []
for k, p in trunk.named_parameters():
    g = p.grad.detach().cpu().double(); r = sd["t." + k].grad.double()
    rows.append((k, ((g - r).norm() / (r.norm() + 1e-30)).item(), r.norm().item()))
for k, e, n in reversed(rows):
    print(f"{k:40s} rel={e:.3e} norm={n:.4g}")
