"""GPU: fixture F2 -- ONE Bottleneck of the reference (koafusion/models/_torchvision.py:83-138) in train mode,
forward AND backward, every tensor in full: block output, BatchNorm buffers after the step, dx and every parameter
gradient, held ELEMENT-WISE at 1e-5 of the tensor's largest magnitude against the imported reference's float32 run
(whose own distance to its float64 run is 2e-7 ... 7e-7, recorded in the fixture).  Three BatchNorms deep there is no
chaotic branch noise (the whole-network gradient bars in test_models_gpu.py are statistical for that reason); the fixture's
input seeds keep every ReLU input at least 2e-5 from zero, so no mask can flip at fp32 rounding level.
Cases: stride 1; stride 2 with the downsample branch; ResNeXt groups 32, both again."""
import numpy as np
import pytest
import torch

import procedural as P
from common import load

pytestmark = pytest.mark.gpu

CASES = (("s1", 256, 64, 1, 1, 64), ("s2ds", 256, 128, 2, 1, 64), ("g32", 256, 64, 1, 32, 4), ("g32s2ds", 256, 128, 2, 32, 4))


def mx(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_bottleneck_f2_elementwise(dev, case):
    from torch import nn
    from oaprogressionmmf_amd.arena import get_arena
    from oaprogressionmmf_amd.models._core_fes import Bottleneck
    from oaprogressionmmf_amd.models._encoder import EncoderFn, _block_fwd
    tag, inpl, planes, stride, groups, bw = case
    g = load("f2_bottleneck.npz")
    seed = int(g[tag + ":seed"])
    N, H, W = 2, 12, 12
    ds = None
    if stride != 1 or inpl != planes * 4:
        ds = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
    blk = Bottleneck(inpl, planes, stride, ds, groups, bw)
    P.fill_state_dict(blk.state_dict())                  # same keys as the reference block: conv1.weight, bn1.*, downsample.0.weight ...
    blk = blk.to(dev)
    buf0 = {k: b.detach().clone() for k, b in blk.named_buffers()}
    get_arena(blk)
    x = torch.relu(torch.from_numpy(P.make_input("f2x_" + tag, (N, inpl, H, W), seed=seed))).to(dev)
    xh = x.permute(0, 2, 3, 1).contiguous()              # NHWC, the product's activation layout
    with torch.no_grad():
        blk.eval()
        r = _block_fwd(blk, xh, N, H, W, False, None)
        ye = r.y.permute(0, 3, 1, 2).cpu().numpy()
        assert mx(ye, g[tag + ":eval"]) < 1e-5, "eval output"
        for k, b in blk.named_buffers():
            assert torch.equal(b, buf0[k]), k               # eval touches no running statistic
        blk.train()
        r = _block_fwd(blk, xh, N, H, W, True, None)
        y = r.y.permute(0, 3, 1, 2).cpu().numpy()
        gy = torch.from_numpy(P.make_input("f2g_" + tag, tuple(y.shape), seed=seed)).to(dev)
        dy = gy.permute(0, 2, 3, 1).contiguous().view(r.y.shape)
        dx = EncoderFn._blocks_bwd([r], dy, None)
        torch.cuda.synchronize()
    errs = {"out": mx(y, g[tag + ":train"]), "dx": mx(dx.view(N, H, W, inpl).permute(0, 3, 1, 2).cpu().numpy(), g[tag + ":dx"])}
    for k, p in blk.named_parameters():
        assert p.grad is not None, k
        errs["grad:" + k] = mx(p.grad.detach().cpu().numpy(), g[f"{tag}:grad:{k}"])
    for k, b in blk.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(g[f"{tag}:buf:{k}"]) == 1
        else:
            errs["buf:" + k] = mx(b.detach().cpu().numpy(), g[f"{tag}:buf:{k}"])
    worst = max(errs, key=errs.get)
    print(f"\n[F2 {tag}] worst element-wise error {errs[worst]:.2e} ({worst}); out {errs['out']:.1e} dx {errs['dx']:.1e}; "
          f"reference fp32 vs its fp64: out/dx {g[tag + ':e32_out_dx'].tolist()}, worst gradient {float(g[tag + ':e32_vals'].max()):.1e}")
    bad = {k: v for k, v in errs.items() if not v < 1e-5}
    assert not bad, bad


def _layer1(dev):
    from oaprogressionmmf_amd.arena import get_arena
    from oaprogressionmmf_amd.models._core_fes import dict_fes
    layer = dict_fes["resnet50"](pretrained=False).layer1        # children named as the reference's: 0.conv1.weight, 0.downsample.0.weight, ...
    P.fill_state_dict(layer.state_dict())
    layer = layer.to(dev)
    get_arena(layer)
    return layer


def _run_layer1(layer, xh, gy_of, N, H, W, rebuild):
    """train-mode forward + backward of the stage through the product's own stage functions; rebuild: the activation-recompute
    path (only the stage input and the BatchNorm statistics survive the forward; the stage is rebuilt before its backward)"""
    from oaprogressionmmf_amd.models._encoder import EncoderFn, _blocks_fwd
    blocks = list(layer.children())
    recs, y, Ho, Wo, _ = _blocks_fwd(blocks, xh, N, H, W, True, slim=rebuild)
    yout = y.clone()
    if rebuild:
        stats = [(r.s1, r.s2, r.s3, r.sd) for r in recs]
        del recs
        recs, y2, _, _, _ = _blocks_fwd(blocks, xh, N, H, W, True, givens=stats)
        assert torch.equal(y2, yout), "rebuilt stage output"
    dy = gy_of(yout)
    dx = EncoderFn._blocks_bwd(recs, dy, None)
    torch.cuda.synchronize()
    return yout, dx


@pytest.mark.parametrize("rebuild", [False, True], ids=["stored", "rebuilt"])
def test_layer1_f2b_elementwise(dev, rebuild):
    """fixture F2b, small: ResNet-50 layer1 of the imported reference (three Bottlenecks, the first with its downsample branch:
    koafusion/models/_torchvision.py:83-138,192-215), train-mode forward + backward, EVERY tensor element-wise at 1e-5 of its
    largest magnitude -- with the bottleneck tails formed in the next conv1's loader, the BatchNorm-backward reductions in the
    dgrad epilogues, the applies in the loaders / plane-image cuts, the halo forward / data gradient, the ring weight gradient,
    and (rebuilt) the activation-recompute path all active as in the full step.  The input seed keeps every ReLU input of the
    float64 run 3.7e-5 from zero (recorded), so no mask flips at fp32 rounding level."""
    g = load("f2b_layer1.npz")
    N, C, H, W = (int(v) for v in g["small:shape"])
    seed = int(g["small:seed"])
    layer = _layer1(dev)
    x = torch.relu(torch.from_numpy(P.make_input("f2bx_small", (N, C, H, W), seed=seed))).to(dev)
    xh = x.permute(0, 2, 3, 1).contiguous()

    def gy_of(y):
        gy = torch.from_numpy(P.make_input("f2bg_small", (N, 256, H, W), seed=seed)).to(dev)
        return gy.permute(0, 2, 3, 1).contiguous().view(y.shape)
    with torch.no_grad():
        y, dx = _run_layer1(layer, xh, gy_of, N, H, W, rebuild)
    errs = {"out": mx(y.view(N, H, W, 256).permute(0, 3, 1, 2).cpu().numpy(), g["small:train"]),
            "dx": mx(dx.view(N, H, W, C).permute(0, 3, 1, 2).cpu().numpy(), g["small:dx"])}
    for k, p in layer.named_parameters():
        assert p.grad is not None, k
        errs["grad:" + k] = mx(p.grad.detach().cpu().numpy(), g["small:grad:" + k])
    for k, b in layer.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(g["small:buf:" + k]) == 1
        else:
            errs["buf:" + k] = mx(b.detach().cpu().numpy(), g["small:buf:" + k])
    worst = max(errs, key=errs.get)
    print(f"\n[F2b small, {'rebuilt' if rebuild else 'stored'}] worst element-wise error {errs[worst]:.2e} ({worst}); out {errs['out']:.1e} "
          f"dx {errs['dx']:.1e}; reference fp32 vs its fp64: out/dx {g['small:e32_out_dx'].tolist()}, worst gradient {float(g['small:e32_vals'].max()):.1e}")
    bad = {k: v for k, v in errs.items() if not v < 1e-5}
    assert not bad, bad


def test_layer1_f2b_multi_tile(dev):
    """fixture F2b, large: the same stage on (4, 64, 48, 48) -- 9216 pixel rows: many tiles of every kernel, the 48-wide rows
    of the halo / ring kernels.  Forward output element-wise (every 29th element) at 1e-5 of the largest magnitude, BatchNorm
    buffers in full at 1e-5.  Gradients against the reference's float64 run: this input is NOT ReLU-safe -- 10.6 M ReLU inputs,
    ~0.8 of them per unit interval around zero, forward values 3e-7 from float64: a handful of masks flip whatever the kernel --
    and one flipped element is one of the 36 864 terms of a per-channel BatchNorm gradient (sum of random-sign terms ~ 192 x one
    term): 5e-3 of that channel.  So: dx and convolution weight gradients (sums over pixels AND channels) norm-wise at 1e-4,
    BatchNorm scale / bias gradients at 1e-3, sampled elements at 1e-2 of the tensor's largest sample (measured 7e-5 / 2.7e-4 /
    2e-3; the element-wise 1e-5 bar on gradients is the ReLU-safe small case above).  Stored and rebuilt stages give the same bits."""
    g = load("f2b_layer1.npz")
    N, C, H, W = (int(v) for v in g["large:shape"])
    layer = _layer1(dev)
    x = torch.relu(torch.from_numpy(P.make_input("f2bx_large", (N, C, H, W), seed=0))).to(dev)
    xh = x.permute(0, 2, 3, 1).contiguous()

    def gy_of(y):
        gy = torch.from_numpy(P.make_input("f2bg_large", (N, 256, H, W), seed=0)).to(dev)
        return gy.permute(0, 2, 3, 1).contiguous().view(y.shape)
    with torch.no_grad():
        y, dx = _run_layer1(layer, xh, gy_of, N, H, W, False)
    yf = y.view(N, H, W, 256).permute(0, 3, 1, 2).contiguous().view(-1)[::29].cpu().numpy()
    assert mx(yf, g["large:train_s29"]) < 1e-5, "forward output"
    for k, b in layer.named_buffers():
        if not k.endswith("num_batches_tracked"):
            assert mx(b.detach().cpu().numpy(), g["large:buf:" + k]) < 1e-5, k
    named = {"dx": dx.view(N, H, W, C).permute(0, 3, 1, 2).cpu().numpy()}
    named.update({"grad:" + k: p.grad.detach().cpu().numpy() for k, p in layer.named_parameters()})
    got = P.summarize_tensors(named, k=32)
    worst_n, worst_s, worst_c, rows = 0.0, 0.0, 0.0, []
    for k in named:
        n64 = float(g[f"large:f64:{k}:norm"])
        s64 = np.asarray(g[f"large:f64:{k}:samples"], np.float64)
        en = abs(float(got[k + ":norm"]) - n64) / n64
        es = float(np.abs(np.asarray(got[k + ":samples"], np.float64) - s64).max() / np.abs(s64).max())
        rows.append((en, es, k))
        worst_n, worst_s = max(worst_n, en), max(worst_s, es)
        if k == "dx" or k.endswith(("conv1.weight", "conv2.weight", "conv3.weight", "downsample.0.weight")):
            worst_c = max(worst_c, en)
    for en, es, k in sorted(rows, reverse=True)[:6]:
        print(f"   {k}: norm {en:.2e} samples {es:.2e}")
    print(f"\n[F2b large] gradient norms vs the reference's float64: worst {worst_n:.2e} (dx / convolution weights {worst_c:.2e}); sampled elements: worst {worst_s:.2e} "
          f"(reference float32 vs float64, L2: median {float(np.median(g['large:e32_vals'])):.1e})")
    assert worst_c < 1e-4 and worst_n < 1e-3 and worst_s < 1e-2, (worst_c, worst_n, worst_s)
    grads = {k: p.grad.detach().clone() for k, p in layer.named_parameters()}
    layer2 = _layer1(dev)
    with torch.no_grad():
        y2, dx2 = _run_layer1(layer2, xh, gy_of, N, H, W, True)
    assert torch.equal(y2, y) and torch.equal(dx2, dx)
    for k, p in layer2.named_parameters():
        assert torch.equal(p.grad, grads[k]), k
