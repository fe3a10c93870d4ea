import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch
from oaprogressionmmf_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for (N, H, W, Cin, Cout, k, s, p) in [(512, 40, 40, 64, 64, 3, 1, 1), (512, 40, 40, 64, 256, 1, 1, 0), (512, 40, 40, 256, 64, 1, 1, 0),
                                      (512, 20, 20, 128, 128, 3, 1, 1), (512, 10, 10, 256, 1024, 1, 1, 0), (200, 5, 5, 512, 2048, 1, 1, 0),
                                      (512, 40, 40, 128, 128, 3, 2, 1)]:
    x = torch.randn(N, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, k, k, Cin, generator=g) * 0.05).to(dev)
    sc = torch.ones(Cin, device=dev); sh = torch.zeros(Cin, device=dev)
    img = ops.build_weight_planes(w, Cout, k * k, Cin)
    img2 = ops.build_weight_planes(w, Cout, k * k, Cin)
    print("planes rebuilt identical:", torch.equal(img[0], img2[0]), torch.equal(img[1], img2[1]), float(img[2]), float(img2[2]))
    noimg = (None, None, img[2])
    for tag, wi in (("images", img), ("in-kernel", noimg)):
        ref = None
        nbad = 0
        for it in range(12):
            y, part = ops.conv2d_fwd(x, w, N, H, W, Cin, Cout, k, k, s, p, sc, sh, stats=True, wimg=wi)
            torch.cuda.synchronize()
            if ref is None:
                ref = (y.clone(), part.clone())
            elif not (torch.equal(ref[0], y) and torch.equal(ref[1], part)):
                nbad += 1
                d = (ref[0] - y).abs()
                if nbad == 1:
                    idx = d.flatten().nonzero().flatten()
                    print("   first diff: count", idx.numel(), "max", float(d.max()), "rows", (idx[:5] // Cout).tolist(), "cols", (idx[:5] % Cout).tolist())
        print(f"conv k{k}s{s} {Cin}->{Cout} px{N*H*W} {tag}: nondeterministic repeats = {nbad}/11")
