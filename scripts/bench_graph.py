"""eager vs HIP-graph replay latency of the inference pass (GPU box): python scripts/bench_graph.py [batch]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch, procedural as P
from oaprogressionmmf_amd.config import ConfigDict
from oaprogressionmmf_amd.models import dict_models
from oaprogressionmmf_amd.run import predict_batch
dev = torch.device("cuda:0")
cfg = P.cfg_xr1mr3c1(dropout=0.1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None).to(dev).eval()
xs = [torch.from_numpy(a).to(dev) for a in P.model_inputs(cfg, B)]
def run():
    return predict_batch(m, xs)
for _ in range(3): lg, pr = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): lg, pr = run()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 20
ref = pr.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): run()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    out = run()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("max |graph - eager| proba:", (out[1] - ref).abs().max().item())
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
graphed = (time.perf_counter() - t0) / 20
print(f"B={B}: eager {eager*1e3:.2f} ms, graph replay {graphed*1e3:.2f} ms")
