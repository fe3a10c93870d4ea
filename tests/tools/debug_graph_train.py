"""bisect helper: capture the train step of a small fusion model into a HIP graph under the stream toggles of the environment"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import procedural as P
from oaprogressionmmf_amd.config import ConfigDict
from oaprogressionmmf_amd.models import dict_models
from oaprogressionmmf_amd.run import GraphedTrainStep
from oaprogressionmmf_amd.various import dict_losses, dict_optimizers
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "full"
cfg = P.cfg_xr1cnn(arch="resnet18", size=160, dropout=0.1) if which == "xr" else P.cfg_full(xr=(96, 96), mr1=(64, 64, 6), mr2=(64, 64, 5), depth=1, dropout=0.1)
m = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None)
P.fill_state_dict(m.state_dict())
m = m.to(dev).train()
B = 2
xs = [torch.from_numpy(a).to(dev) for a in P.model_inputs(cfg, B, 11)]
ys = torch.from_numpy(P.make_target("target", B, 11)).to(dev)
loss_fn = dict_losses["FocalLoss"](reduction="mean", gamma=2.0, num_classes=2)
opt = dict_optimizers["Adam"](m.parameters(), lr=1e-3, weight_decay=1e-4, capturable=True)
step = GraphedTrainStep(m, loss_fn, opt, xs, ys, warmup=1, seed=1)
for it in range(4):
    lg, ls = step(xs, ys)
    print(it, float(ls), flush=True)
print("ok", which, {k: os.environ.get(k) for k in ("KOAF_ENCODER_LANES", "KOAF_SIDE_STREAM")})
