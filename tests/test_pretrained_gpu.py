"""GPU: the convolution kernels on an ImageNet-checkpoint-LIKE parameter distribution.

Every shipped reference recipe builds its encoders with `pretrained: true` (runner.sh:94,116,132,148,169; koafusion/models/
_torchvision.py:249-261).  The checkpoint files cannot travel here, so this test draws parameters from the distribution of a
trained ResNet (procedural.imagenet_like_fill: BatchNorm scale 1e-3 .. 3 with exact zeros and negative entries, running variance
1e-4 .. 10, heavy-tailed convolution weights with per-layer gains over two decades) and holds the HIP trunk to the float64
oracle on them.  What it protects: the fp16-piece scheme's FIXED activation scale (koaf.h KOAF_ACT_SCALE = 16: |x| <= 4094 behind
a BatchNorm) and the amax-derived weight / gradient scales, which the fixtures' gentle fill (scale 1 +- 0.1) never stressed.
Bars: whole trunks on two images -- train / eval outputs within BASELINE's 1e-3 of float64 (measured 6e-5 ResNet-50, 4e-4
ResNeXt-50: on this law the map itself is badly conditioned -- 50 pixels per layer4 channel -- the oracle's own float32 run sits
6e-6 ... 8e-6 from float64, 50 x its distance on the fixtures' weights), numerics status words (0, 0), BatchNorm buffers 2e-4,
gradients at plumbing level; ONE Bottleneck (dense, strided + downsample, grouped) element-wise at 2e-4: the arithmetic bar on
this law (output and dx measured at 1e-5)."""
import numpy as np
import pytest
import torch

import procedural as P
from common import rel

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("arch,shape", [("resnet50", (2, 1, 160, 160)), ("resnext50_32x4d", (2, 1, 130, 130))])
def test_trunk_on_imagenet_like_weights(dev, arch, shape):
    from oracle import koafusion_cpu as O
    from oaprogressionmmf_amd import ops
    from oaprogressionmmf_amd.models._core_fes import dict_fes
    from oaprogressionmmf_amd.models._encoder import KoafTrunk
    net = dict_fes[arch](pretrained=False)
    trunk = KoafTrunk(*list(net.children())[:-1])
    P.fill_state_dict(trunk.state_dict(), fill=P.imagenet_like_fill)
    trunk = trunk.to(dev)
    for m in trunk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0          # running statistics := this batch's (self-consistent like a trained checkpoint's), for the eval leg
    x = t(P.make_input("inet", shape)).to(dev)
    gy = None
    ops.numerics_status(reset=True)

    def oracle(dtype):
        spec = O.trunk_spec("t", arch)
        sd = {k: t(P.imagenet_like_fill(k[2:], s, dt == torch.int64)).reshape(s) for k, s, dt in spec}
        sd = {k: (v if v.dtype == torch.int64 else v.to(dtype)) for k, v in sd.items()}
        for k in sd:
            if O.is_param(k):
                sd[k].requires_grad_(True)
        yo = O.trunk(x.cpu().to(dtype), sd, "t", arch, True)
        (yo * gy.to(dtype)).sum().backward()
        return sd, yo.detach()
    trunk.train()
    y = trunk(x)
    gy = t(P.make_input("inetg", tuple(y.shape)))
    (y * gy.to(dev)).sum().backward()
    sd64, y64 = oracle(torch.float64)
    sd32, y32 = oracle(torch.float32)
    e_out = rel(y.detach().cpu().numpy(), y64.numpy())
    print(f"\n[{arch}] imagenet-like weights: train output vs float64 {e_out:.2e} (oracle float32: {rel(y32.numpy(), y64.numpy()):.2e})")
    assert e_out < (2e-4 if arch == "resnet50" else 1e-3), "train output"     # (ResNeXt: measured 3.9e-4, see the module docstring)
    # Whole-trunk gradients on TWO images are deep in the branch-noise regime on these weights (the oracle's own float32 run sits
    # 5e-3 (median) from its float64 run: a ReLU mask that flips in layer4 is one of 50 pixels of its channel): plumbing-level bar
    # here -- every gradient finite, the median within 0.1 -- and the tight, element-wise gradient bar on this parameter law
    # is held where no mask can flip: test_bottleneck_on_imagenet_like_weights below.
    truth = {k[2:]: v.grad.numpy() for k, v in sd64.items() if O.is_param(k)}
    noise = {k[2:]: rel(v.grad.numpy(), truth[k[2:]]) for k, v in sd32.items() if O.is_param(k)}
    mine = {k: p.grad.cpu().numpy() for k, p in trunk.named_parameters()}
    errs = {k: rel(mine[k], truth[k]) for k in truth}
    print(f"[{arch}] gradients vs float64: median {np.median(list(errs.values())):.2e}, worst {max(errs.values()):.2e} "
          f"(oracle float32: median {np.median(list(noise.values())):.2e}, worst {max(noise.values()):.2e})")
    assert all(np.isfinite(v).all() for v in mine.values())
    assert np.median(list(errs.values())) < 0.1
    # the zero-scale channels really carry no gradient into their convolution... (the BatchNorm scale itself still gets one)
    st = ops.numerics_status()
    assert (st["saturated"], st["nonfinite"]) == (0, 0), st
    # eval leg on the statistics the train pass left (momentum 1: the batch's own)
    bufs = {k: b.detach().cpu() for k, b in trunk.named_buffers()}
    for k, b in bufs.items():
        if k.endswith(("running_mean", "running_var")):
            ref = sd64["t." + k]
            # (the oracle's own update ran with momentum 0.1 from the same start: undo it to get the batch statistic)
            start = t(P.imagenet_like_fill(k, tuple(b.shape))).double()
            batch = (ref - 0.9 * start) / 0.1
            assert rel(b.numpy(), batch.numpy()) < 1e-3, k
    trunk.eval()
    with torch.no_grad():
        ye = trunk(x)
    sde = {k: v.detach().clone() for k, v in sd64.items()}
    for k, b in bufs.items():
        sde["t." + k] = b.double() if b.dtype.is_floating_point else b
    with torch.no_grad():
        ye64 = O.trunk(x.cpu().double(), sde, "t", arch, False)
    e_eval = rel(ye.cpu().numpy(), ye64.numpy())
    print(f"[{arch}] eval output vs float64 {e_eval:.2e}")
    assert e_eval < (2e-4 if arch == "resnet50" else 1e-3), "eval output"
    st = ops.numerics_status()
    assert (st["saturated"], st["nonfinite"]) == (0, 0), st


def _bottleneck_preacts(sd, x, stride, has_ds, groups=1):
    """the three ReLU inputs of one Bottleneck in float64 on the CPU (train-mode BatchNorm), for the margin search"""
    import torch.nn.functional as F

    def bn(v, p):
        return F.batch_norm(v, None, None, sd[p + ".weight"], sd[p + ".bias"], training=True, eps=1e-5)
    a1 = bn(F.conv2d(x, sd["conv1.weight"]), "bn1")
    a2 = bn(F.conv2d(torch.relu(a1), sd["conv2.weight"], stride=stride, padding=1, groups=groups), "bn2")
    idt = bn(F.conv2d(x, sd["downsample.0.weight"], stride=stride), "downsample.1") if has_ds else x
    a3 = bn(F.conv2d(torch.relu(a2), sd["conv3.weight"]), "bn3") + idt
    return a1, a2, a3


@pytest.mark.parametrize("tag,inpl,planes,stride,groups,bw", [("s1", 256, 64, 1, 1, 64), ("s2ds", 256, 128, 2, 1, 64), ("g32", 256, 64, 1, 32, 4)])
def test_bottleneck_on_imagenet_like_weights(dev, tag, inpl, planes, stride, groups, bw):
    """ONE Bottleneck (koafusion/models/_torchvision.py:83-138) with parameters from the ImageNet-checkpoint-like law -- BatchNorm
    scales from 1e-3 to 3, exact zeros, negative entries, heavy-tailed convolution weights over two decades of gain -- train-mode
    forward and backward, EVERY tensor element-wise against the float64 oracle at 2e-4 of its largest magnitude (measured: output
    1.1e-5, dx 8.8e-6, parameter gradients <= 1e-5 but one BatchNorm scale gradient at 9e-5).  The input seed is searched
    (float64, CPU) for the largest ReLU margin available (2e-5); on this law pre-activations reach ~10 and the forward sits 1e-5
    of that from float64 -- ten times the fixtures' law, the price of ONE scale per tensor over channels six decades apart -- so
    the margin does not exclude every flip and the bar is 10 x F2's.  What it holds: the fixed activation scale, the
    amax-derived weight / gradient scales and the BatchNorm-backward apply formed on load stay at this level on such channels,
    with no saturation (status words 0 / 0)."""
    from torch import nn
    from oracle import koafusion_cpu as O
    from oaprogressionmmf_amd import ops
    from oaprogressionmmf_amd.arena import get_arena
    from oaprogressionmmf_amd.models._core_fes import Bottleneck
    from oaprogressionmmf_amd.models._encoder import EncoderFn, _block_fwd
    N, H, W = 2, 12, 12
    has_ds = stride != 1 or inpl != planes * 4
    ds = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4)) if has_ds else None
    blk = Bottleneck(inpl, planes, stride, ds, groups, bw)
    P.fill_state_dict(blk.state_dict(), fill=P.imagenet_like_fill)
    sd64 = {k: v.detach().double().clone() for k, v in blk.state_dict().items()}
    seed = None
    for cand in range(400):
        x64 = torch.relu(t(P.make_input("inetb_" + tag, (N, inpl, H, W), seed=cand))).double()
        with torch.no_grad():
            margin = min(float(a.abs().min()) for a in _bottleneck_preacts(sd64, x64, stride, has_ds, groups))
        if margin > 2e-5:
            seed = cand
            break
    assert seed is not None
    x = torch.relu(t(P.make_input("inetb_" + tag, (N, inpl, H, W), seed=seed)))
    # float64 truth through the oracle's Bottleneck
    sdo = {"b." + k: v.clone() for k, v in sd64.items()}
    for k in sdo:
        if O.is_param(k):
            sdo[k].requires_grad_(True)
    xo = x.double().requires_grad_(True)
    yo = O._bottleneck(xo + 0, sdo, "b", True, stride, groups)
    gy = t(P.make_input("inetbg_" + tag, tuple(yo.shape), seed=seed))
    (yo * gy.double()).sum().backward()
    blk = blk.to(dev).train()
    get_arena(blk)
    ops.numerics_status(reset=True)
    xh = x.to(dev).permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        r = _block_fwd(blk, xh, N, H, W, True, None)
        y = r.y.permute(0, 3, 1, 2).cpu()
        dy = gy.to(dev).permute(0, 2, 3, 1).contiguous().view(r.y.shape)
        dx = EncoderFn._blocks_bwd([r], dy, None)
        torch.cuda.synchronize()

    def mx(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
    errs = {"out": mx(y.numpy(), yo.detach().numpy()), "dx": mx(dx.view(N, H, W, inpl).permute(0, 3, 1, 2).cpu().numpy(), xo.grad.numpy())}
    for k, p in blk.named_parameters():
        errs["grad:" + k] = mx(p.grad.detach().cpu().numpy(), sdo["b." + k].grad.numpy())
    worst = max(errs, key=errs.get)
    print(f"\n[imagenet-like Bottleneck {tag}] seed {seed}, ReLU margin {margin:.1e}: worst element-wise error {errs[worst]:.2e} ({worst}); "
          f"out {errs['out']:.1e} dx {errs['dx']:.1e}")
    assert not {k: v for k, v in errs.items() if not v < 2e-4}, errs
    st = ops.numerics_status()
    assert (st["saturated"], st["nonfinite"]) == (0, 0), st
