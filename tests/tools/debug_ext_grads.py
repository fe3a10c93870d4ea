"""per-parameter gradient error of an extension model vs the fp64 oracle (GPU box)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import procedural as P
from common import rel
from test_models_gpu import build, t
from test_ext_gpu import CASES
from oracle import koafusion_cpu as O
from oaprogressionmmf_amd.various import dict_losses

case = sys.argv[1] if len(sys.argv) > 1 else "XR1C1Cnn"
dev = torch.device("cuda:0")
if case.startswith("xr1cnn"):
    _, arch, Bs, size = case.split(":")
    cfg, B = P.cfg_xr1cnn(arch=arch, size=int(size)), int(Bs)
else:
    cfg, B = CASES[case]()
xs = [t(a) for a in P.model_inputs(cfg, B, 42)]
y = t(P.make_target("target", B, 42))
m = build(cfg, dev).train()
o32 = O.OracleModel(cfg, fill=P.fill_value)
o64 = O.OracleModel(cfg, fill=P.fill_value, dtype=torch.float64)
loss_fn = dict_losses["FocalLoss"](reduction="mean", gamma=2.0, num_classes=2)
lg = m(*[x.to(dev) for x in xs])["main"]
loss = loss_fn(input=lg.squeeze(1), target=y.to(dev).long().squeeze(1))
loss.backward()
o32.train_step(xs, y, optimize=False)
o64.train_step(xs, y, optimize=False)
truth = {k: p.grad.numpy() for k, p in o64.named_parameters() if p.grad is not None}
n32 = {k: rel(p.grad.numpy(), truth[k]) for k, p in o32.named_parameters() if p.grad is not None}
mine = {k: rel(p.grad.detach().cpu().numpy(), truth[k]) for k, p in m.named_parameters() if p.grad is not None}
worst = max(mine[k] / (n32[k] + 1e-4) for k in truth)
print(case, "worst ratio", worst)
for k in (truth if worst > 5 and "-v" in sys.argv else []):
    print(f"{k:50s} |g|={np.linalg.norm(truth[k]):.3e} mine={mine[k]:.2e} cpu32={n32[k]:.2e} ratio={mine[k]/(n32[k]+1e-4):.1f}")
