"""GPU: the convolution kernels on an ImageNet-checkpoint-LIKE parameter distribution.

Every shipped reference recipe builds its encoders with `pretrained: true` (runner.sh:94,116,132,148,169; koafusion/models/
_torchvision.py:249-261).  The checkpoint files cannot travel here, so this test draws parameters from the distribution of a
trained ResNet (procedural.imagenet_like_fill: BatchNorm scale 1e-3 .. 3 with exact zeros and negative entries, running variance
1e-4 .. 10, heavy-tailed convolution weights with per-layer gains over two decades) and holds the HIP trunk to the float64
oracle on them.  What it protects: the fp16-piece scheme's FIXED activation scale (koaf.h KOAF_ACT_SCALE = 16: |x| <= 4094 behind
a BatchNorm) and the amax-derived weight / gradient scales, which the fixtures' gentle fill (scale 1 +- 0.1) never stressed.
Bars: train / eval outputs within 2e-4 of float64 (BASELINE: 1e-3), numerics status words (0, 0), gradients against float64
no worse than 10 x the oracle's own float32 run on the same graph (floor 1e-4), BatchNorm buffers 2e-4."""
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
    assert e_out < 2e-4, "train output"
    truth = {k[2:]: v.grad.numpy() for k, v in sd64.items() if O.is_param(k)}
    noise = {k[2:]: rel(v.grad.numpy(), truth[k[2:]]) for k, v in sd32.items() if O.is_param(k)}
    mine = {k: p.grad.cpu().numpy() for k, p in trunk.named_parameters()}
    errs = {k: rel(mine[k], truth[k]) for k in truth}
    ratio = {k: errs[k] / (noise[k] + 1e-4) for k in truth}
    wk = max(ratio, key=ratio.get)
    print(f"[{arch}] gradients vs float64: median {np.median(list(errs.values())):.2e} (oracle float32 {np.median(list(noise.values())):.2e}), "
          f"worst ratio {ratio[wk]:.1f} at {wk} ({errs[wk]:.2e} vs {noise[wk]:.2e})")
    assert np.median(list(ratio.values())) <= 2.0 and ratio[wk] <= 10.0, (wk, errs[wk], noise[wk])
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
            assert rel(b.numpy(), batch.numpy()) < 2e-4, k
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
    assert e_eval < 2e-4, "eval output"
    st = ops.numerics_status()
    assert (st["saturated"], st["nonfinite"]) == (0, 0), st
