#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE (imported from
/root/reference, read-only) on CPU with procedural weights/inputs (tests/procedural.py).

Only DATA leaves this script (.npz / .json of inputs' seeds and expected outputs); no reference source,
bytecode or pickle is written into the repo.  Import recipe: torchvision is absent, so
`torchvision.models.resnet18/34/50` are served from the reference's own verbatim copy
koafusion/models/_torchvision.py; _losses.py / _optimizers.py are loaded by path (the package
__init__ of koafusion.various imports cv2/nibabel, which are absent).

Usage:  python tests/golden/make_golden.py [case ...]      (no args = all cases)
"""
import importlib.util
import json
import sys
import time
import types
from pathlib import Path

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import procedural as P  # noqa: E402

REF = Path("/root/reference")


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    tv = _load_by_path("_ref_torchvision", REF / "koafusion/models/_torchvision.py")
    fake = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")

    def _absent(name):
        def f(*a, **k):
            raise RuntimeError(f"torchvision.models.{name} is not available in this container")
        return f
    for n in ("squeezenet1_0", "vgg16", "densenet161", "inception_v3"):
        setattr(models, n, _absent(n))
    models.resnet18, models.resnet34, models.resnet50 = tv.resnet18, tv.resnet34, tv.resnet50
    fake.models = models
    sys.modules["torchvision"] = fake
    sys.modules["torchvision.models"] = models
    sys.path.insert(0, str(REF))
    import koafusion.models as km
    from koafusion import preproc
    losses = _load_by_path("_ref_losses", REF / "koafusion/various/_losses.py")
    optims = _load_by_path("_ref_optims", REF / "koafusion/various/_optimizers.py")
    return km, tv, preproc, losses, optims


class Cfg(dict):
    """attr + item access, like OmegaConf's DictConfig"""

    def __init__(self, d):
        super().__init__()
        for k, v in d.items():
            self[k] = Cfg(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def set_dropout_zero_train(model):
    model.train()


def grads_summary(model):
    named, none = {}, []
    for k, p in model.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            named["grad:" + k] = p.grad.detach().numpy()
    out = P.summarize_tensors(named)
    out["none_grad_keys"] = np.array(none)
    return out


def buffers_summary(model):
    named = {}
    for k, b in model.named_buffers():
        named["buf:" + k] = b.detach().numpy()
    return P.summarize_tensors(named)


def params_summary(model):
    return P.summarize_tensors({"param:" + k: p.detach().numpy() for k, p in model.named_parameters()})


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def fp64_twin(build, run, grads32, logits32):
    """Run the same reference computation in float64 and record, per parameter, how far the reference's OWN
    fp32 gradients are from it (`e32`).  Through ~50 train-mode BatchNorm layers fp32 rounding (ReLU-mask
    flips, small-batch statistics) moves gradients by up to a few 1e-3 relative, so parity of a different
    fp32 implementation is stated against this noise floor, not against bit patterns."""
    m64 = build().double()
    logits64, grads64 = run(m64, torch.float64)
    keys = [k for k in grads32 if k in grads64]
    out = {"e32_keys": np.array(keys), "e32_vals": np.array([_rel(grads32[k], grads64[k]) for k in keys]),
           "e32_logits": np.float64(_rel(logits32, logits64)), "train_logits64": logits64.numpy()}
    out.update(P.summarize_tensors({"g64:" + k: grads64[k].numpy() for k in keys}))
    return out


def run_model_case(km, losses, cfg, B, fname, eval_fwd=True, train_step=True, adam_steps=0, seed=1234, tensor_shapes=None,
                   twin=True):
    """tensor_shapes: sizes of the input TENSORS when they differ from the config's `input_size` (the reference's
    constructors assert on config sizes only; forward never checks tensor shapes); twin=False skips the float64 run
    (full-size cases whose saved activations would not fit the build container in float64)."""
    t0 = time.time()

    def build():
        torch.manual_seed(0)
        m = km.dict_models[cfg["name"]](config=Cfg(cfg), path_weights=None)
        P.fill_state_dict(m.state_dict())
        return m

    model = build()
    xs = [t(a) for a in P.model_inputs(dict(cfg, input_size=tensor_shapes) if tensor_shapes else cfg, B, seed)]
    y = t(P.make_target("target", B, seed))
    loss_fn = losses.FocalLoss(reduction="mean", gamma=2.0, num_classes=2)
    out = {"B": np.int64(B), "cfg_json": np.array(json.dumps(cfg)), "seed": np.int64(seed),
           "torch_version": np.array(torch.__version__)}
    if tensor_shapes:
        out["tensor_shapes_json"] = np.array(json.dumps(tensor_shapes))
    if eval_fwd:
        model.eval()
        with torch.no_grad():
            out["eval_logits"] = model(*xs)["main"].numpy()
    if train_step:
        model.train()   # all dropout p = 0 in the fixture configs; BN in batch-stat mode
        logits = model(*xs)["main"]
        loss = loss_fn(input=logits.squeeze(1), target=y.long().squeeze(1))
        loss.backward()
        out["train_logits"] = logits.detach().numpy()
        out["train_loss"] = np.float64(loss.item())
        out.update(grads_summary(model))
        out.update(buffers_summary(model))
        nbt = [b.item() for k, b in model.named_buffers() if k.endswith("num_batches_tracked")]
        out["num_batches_tracked"] = np.array(nbt, dtype=np.int64)

        def run64(m64, dt):
            m64.train()
            lg = m64(*[x.to(dt) for x in xs])["main"]
            ls = loss_fn(input=lg.squeeze(1), target=y.long().squeeze(1))
            ls.backward()
            return lg.detach(), {k: p.grad.detach() for k, p in m64.named_parameters() if p.grad is not None}
        if twin:
            g32 = {k: p.grad.detach() for k, p in model.named_parameters() if p.grad is not None}
            out.update(fp64_twin(build, run64, g32, logits.detach()))
    if adam_steps:
        # continue from the state after the train step above: 3 Adam steps on the same batch
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4)
        ls = []
        for s in range(adam_steps):
            if s > 0 or not train_step:
                opt.zero_grad()
                logits = model(*xs)["main"]
                loss = loss_fn(input=logits.squeeze(1), target=y.long().squeeze(1))
                loss.backward()
            ls.append(loss.item())
            opt.step()
        out["adam_losses"] = np.array(ls, dtype=np.float64)
        ps = params_summary(model)
        out.update({"adam:" + k: v for k, v in ps.items()})
    np.savez_compressed(HERE / fname, **out)
    print(f"  wrote {fname} ({(HERE / fname).stat().st_size / 1024:.0f} KiB) in {time.time() - t0:.1f}s")


# ------------------------------------------------------------------------------------------------
def case_f1_attention_feat(km, **_):
    out = {}
    for dim, heads, n in ((64, 4, 25), (2048, 8, 12)):
        torch.manual_seed(0)
        att = km.Attention(dim, heads=heads, dropout=0.0)
        P.fill_state_dict(att.state_dict())
        x = t(P.make_input(f"att{dim}", (2, n, dim))).requires_grad_(True)
        o, a = att(x)
        (o * t(P.make_input(f"attg{dim}", (2, n, dim)))).sum().backward()
        out[f"att{dim}:out"] = o.detach().numpy()
        out[f"att{dim}:attn"] = a.detach().numpy()
        out[f"att{dim}:dx"] = x.grad.numpy()
        out[f"att{dim}:dwqkv_norm"] = np.float64(att.to_qkv.weight.grad.double().norm().item())
    for with_cls in (True, False):
        torch.manual_seed(0)
        f = km.FeaT(num_patches=25, patch_dim=64, emb_dim=64, depth=2, heads=4, mlp_dim=128, num_classes=2,
                    with_cls=with_cls)
        P.fill_state_dict(f.state_dict())
        f.eval()
        x = t(P.make_input("feat", (3, 25, 64)))
        o, s, a = f(x)
        tag = f"feat_cls{int(with_cls)}"
        out[tag + ":outputs"] = o.detach().numpy()
        out[tag + ":states"] = s.detach().numpy()
        out[tag + ":attn0"] = a[0].detach().numpy()
    np.savez_compressed(HERE / "f1_attention_feat.npz", **out)
    print("  wrote f1_attention_feat.npz")


F2_CASES = (                      # tag, inplanes, planes, stride, groups, base_width, (N, H, W)
    ("s1", 256, 64, 1, 1, 64, (2, 12, 12)),
    ("s2ds", 256, 128, 2, 1, 64, (2, 12, 12)),
    ("g32", 256, 64, 1, 32, 4, (2, 12, 12)),
    ("g32s2ds", 256, 128, 2, 32, 4, (2, 12, 12)),
)


def case_f2_bottleneck(tv, **_):
    """F2: ONE `Bottleneck` (_torchvision.py:83-138) in train mode, forward + backward, FULL tensors -- output, BatchNorm
    buffers after the step, dx and every parameter gradient -- plus the eval-mode output.  Three BatchNorms deep there is no
    chaotic branch noise, so this is where a tight element-wise bar on gradients is possible; the only discrete effect left
    is a ReLU input that close to zero that a 1e-6 rounding difference takes the other branch, so the input seed is chosen
    (first of 0, 1, 2, ...) such that no ReLU input of the float64 run lies within 2e-5 of zero (the margin is recorded)."""
    out = {"torch_version": np.array(torch.__version__)}
    for tag, inpl, planes, stride, groups, bw, (N, H, W) in F2_CASES:
        def build(dt=torch.float32):
            torch.manual_seed(0)
            ds = None
            if stride != 1 or inpl != planes * 4:
                ds = torch.nn.Sequential(tv.conv1x1(inpl, planes * 4, stride), torch.nn.BatchNorm2d(planes * 4))
            blk = tv.Bottleneck(inpl, planes, stride=stride, downsample=ds, groups=groups, base_width=bw)
            P.fill_state_dict(blk.state_dict())
            return blk.to(dt)

        def preacts(blk, x):
            """the three ReLU inputs of Bottleneck.forward, restated on the block's own children (float64 margin check only)"""
            a1 = blk.bn1(blk.conv1(x))
            a2 = blk.bn2(blk.conv2(torch.relu(a1)))
            a3 = blk.bn3(blk.conv3(torch.relu(a2))) + (blk.downsample(x) if blk.downsample is not None else x)
            return a1, a2, a3
        seed = None
        for cand in range(200):
            x64 = torch.relu(t(P.make_input("f2x_" + tag, (N, inpl, H, W), seed=cand))).double()
            b64 = build(torch.float64).train()
            with torch.no_grad():
                margin = min(float(a.abs().min()) for a in preacts(b64, x64))
            if margin > 2e-5:
                seed = cand
                break
        assert seed is not None, tag
        x = torch.relu(t(P.make_input("f2x_" + tag, (N, inpl, H, W), seed=seed)))
        blk = build()
        blk.eval()
        with torch.no_grad():
            out[tag + ":eval"] = blk(x.clone()).numpy()
        res = {}
        for dt in (torch.float32, torch.float64):
            b = build(dt).train()
            xi = x.to(dt).clone().requires_grad_(True)
            y = b(xi + 0)                              # (+0: the block's `out += identity` must not write into the leaf)
            g = t(P.make_input("f2g_" + tag, tuple(y.shape), seed=seed)).to(dt)
            (y * g).sum().backward()
            res[dt] = (y.detach(), xi.grad.detach(), {k: p.grad.detach() for k, p in b.named_parameters()},
                       {k: v.detach().clone() for k, v in b.named_buffers()})
        y32, dx32, g32, buf32 = res[torch.float32]
        y64, dx64, g64, _ = res[torch.float64]
        out[tag + ":seed"] = np.int64(seed)
        out[tag + ":relu_margin64"] = np.float64(margin)
        out[tag + ":train"] = y32.numpy()
        out[tag + ":dx"] = dx32.numpy()
        for k, v in g32.items():
            out[tag + ":grad:" + k] = v.numpy()
        for k, v in buf32.items():
            out[tag + ":buf:" + k] = v.numpy()
        keys = sorted(g32)
        e32 = [float((g32[k].double() - g64[k]).abs().max() / g64[k].abs().max()) for k in keys]
        # how far the reference's own float32 results are from its float64 run (max-norm, relative to the tensor's largest
        # magnitude): the noise floor under the 1e-5 bar of the tests
        out[tag + ":e32_keys"] = np.array(keys)
        out[tag + ":e32_vals"] = np.array(e32)
        out[tag + ":e32_out_dx"] = np.array([float((y32.double() - y64).abs().max() / y64.abs().max()),
                                             float((dx32.double() - dx64).abs().max() / dx64.abs().max())])
        print(f"  {tag}: seed {seed}, ReLU margin {margin:.2e}, reference fp32 vs fp64: out {_rel(y32, y64):.1e} dx "
              f"{float((dx32.double() - dx64).abs().max() / dx64.abs().max()):.1e} worst grad (max-norm) {max(e32):.1e}")
    np.savez_compressed(HERE / "f2_bottleneck.npz", **out)
    print(f"  wrote f2_bottleneck.npz ({(HERE / 'f2_bottleneck.npz').stat().st_size / 1024:.0f} KiB)")


F2B_SIZES = (("small", (2, 8, 8)), ("large", (4, 48, 48)))


def case_f2b_stage(tv, **_):
    """F2b: ResNet-50 `layer1` of the reference (three Bottlenecks, the first with its downsample branch: _torchvision.py:83-138,
    192-215) in train mode, forward + backward -- the rung between F2 (one block, element-wise) and F3 / F14 (whole trunks, at
    branch-noise level).  Two sizes:
      small (2, 64, 8, 8): EVERY tensor in full (output, dx, every parameter gradient, BatchNorm buffers), with an input seed
        such that no ReLU input of the float64 run lies within 1e-5 of zero -- element-wise bars are meaningful only where no
        mask can flip at fp32 rounding level, and the nine ReLU layers see 1152 values per pixel: 147 k values here (a
        (4, 64, 48, 48) input has 10.6 M of them, ~170 of which sit within 2e-5 of zero whatever the seed);
      large (4, 64, 48, 48): the forward output (every 29th element: flips move it by <= 1e-6), BatchNorm buffers in full, and
        per-tensor gradient norms + samples with the reference's own float32-vs-float64 noise beside them (e32), for the
        multi-tile paths of the same kernels."""
    out = {"torch_version": np.array(torch.__version__)}

    def build(dt=torch.float32):
        torch.manual_seed(0)
        layer = tv.resnet50().layer1
        P.fill_state_dict(layer.state_dict())
        return layer.to(dt)

    def margin_of(layer, x):
        m = float("inf")
        for blk in layer:
            a1 = blk.bn1(blk.conv1(x))
            a2 = blk.bn2(blk.conv2(torch.relu(a1)))
            a3 = blk.bn3(blk.conv3(torch.relu(a2))) + (blk.downsample(x) if blk.downsample is not None else x)
            m = min(m, float(a1.abs().min()), float(a2.abs().min()), float(a3.abs().min()))
            x = torch.relu(a3)
        return m
    for tag, (N, H, W) in F2B_SIZES:
        seed, margin = 0, None
        if tag == "small":
            seed = None
            for cand in range(400):
                x64 = torch.relu(t(P.make_input("f2bx_" + tag, (N, 64, H, W), seed=cand))).double()
                with torch.no_grad():
                    margin = margin_of(build(torch.float64).train(), x64)
                if margin > 1e-5:
                    seed = cand
                    break
            assert seed is not None
        x = torch.relu(t(P.make_input("f2bx_" + tag, (N, 64, H, W), seed=seed)))
        res = {}
        for dt in (torch.float32, torch.float64):
            layer = build(dt).train()
            xi = x.to(dt).clone().requires_grad_(True)
            y = layer(xi + 0)
            g = t(P.make_input("f2bg_" + tag, tuple(y.shape), seed=seed)).to(dt)
            (y * g).sum().backward()
            res[dt] = (y.detach(), xi.grad.detach(), {k: p.grad.detach() for k, p in layer.named_parameters()},
                       {k: v.detach().clone() for k, v in layer.named_buffers()})
        y32, dx32, g32, buf32 = res[torch.float32]
        y64, dx64, g64, _ = res[torch.float64]
        out[tag + ":seed"] = np.int64(seed)
        out[tag + ":shape"] = np.array([N, 64, H, W])
        keys = sorted(g32)
        out[tag + ":e32_keys"] = np.array(keys)
        for k, v in buf32.items():
            out[tag + ":buf:" + k] = v.numpy()
        if tag == "small":
            out[tag + ":relu_margin64"] = np.float64(margin)
            out[tag + ":train"] = y32.numpy()
            out[tag + ":dx"] = dx32.numpy()
            for k, v in g32.items():
                out[tag + ":grad:" + k] = v.numpy()
            e32 = [float((g32[k].double() - g64[k]).abs().max() / g64[k].abs().max()) for k in keys]
            out[tag + ":e32_vals"] = np.array(e32)
            out[tag + ":e32_out_dx"] = np.array([float((y32.double() - y64).abs().max() / y64.abs().max()),
                                                 float((dx32.double() - dx64).abs().max() / dx64.abs().max())])
            print(f"  {tag}: seed {seed}, ReLU margin {margin:.2e}, reference fp32 vs fp64 (max-norm): out {out[tag + ':e32_out_dx'][0]:.1e} "
                  f"dx {out[tag + ':e32_out_dx'][1]:.1e} worst grad {max(e32):.1e}")
        else:
            out[tag + ":train_s29"] = y32.numpy().reshape(-1)[::29].copy()
            out[tag + ":train64_s29"] = y64.numpy().reshape(-1)[::29].copy()
            named = {"dx": dx32.numpy()}
            named.update({"grad:" + k: v.numpy() for k, v in g32.items()})
            out.update({tag + ":" + k: v for k, v in P.summarize_tensors(named, k=32).items()})
            named64 = {"dx": dx64.numpy()}
            named64.update({"grad:" + k: v.numpy() for k, v in g64.items()})
            out.update({tag + ":f64:" + k: v for k, v in P.summarize_tensors(named64, k=32).items()})
            e32 = [_rel(g32[k], g64[k]) for k in keys]
            out[tag + ":e32_vals"] = np.array(e32)           # relative L2 here
            out[tag + ":e32_dx"] = np.float64(_rel(dx32, dx64))
            print(f"  {tag}: reference fp32 vs fp64 (L2): out {_rel(y32, y64):.1e} dx {_rel(dx32, dx64):.1e} grads median "
                  f"{np.median(e32):.1e} worst {max(e32):.1e}")
    np.savez_compressed(HERE / "f2b_layer1.npz", **out)
    print(f"  wrote f2b_layer1.npz ({(HERE / 'f2b_layer1.npz').stat().st_size / 1024:.0f} KiB)")


def case_f3_trunk(tv, **_):
    out = {}
    for arch, shape in (("resnet50", (4, 1, 160, 160)), ("resnet50", (2, 1, 96, 112)), ("resnext50_32x4d", (2, 1, 130, 130)),
                        ("resnet18", (2, 1, 96, 96)), ("resnet34", (2, 1, 64, 96))):
        def build():
            torch.manual_seed(0)
            net = getattr(tv, arch)(pretrained=False)
            tr = torch.nn.Sequential(*list(net.children())[:-1])
            P.fill_state_dict(tr.state_dict())
            return tr
        trunk = build()
        x = t(P.make_input("trunk", shape)).repeat(1, 3, 1, 1)
        tag = f"{arch}_{shape[0]}x{shape[2]}x{shape[3]}"
        trunk.eval()
        with torch.no_grad():
            out[tag + ":eval"] = trunk(x).numpy()

        def run(m, dt):
            m.train()
            yy = m(x.to(dt))
            (yy * t(P.make_input("trunkg", tuple(yy.shape))).to(dt)).sum().backward()
            return yy.detach(), {k: p.grad.detach() for k, p in m.named_parameters()}
        y, g32 = run(trunk, torch.float32)
        out[tag + ":train"] = y.numpy()
        out.update({tag + ":" + k: v for k, v in P.summarize_tensors(
            {"grad:" + k: v.numpy() for k, v in g32.items()}).items()})
        out.update({tag + ":" + k: v for k, v in P.summarize_tensors(
            {"buf:" + k: b.numpy() for k, b in trunk.named_buffers()}).items()})
        out.update({tag + ":" + k: v for k, v in fp64_twin(build, run, g32, y).items()})
    np.savez_compressed(HERE / "f3_trunk.npz", **out)
    print("  wrote f3_trunk.npz")


def case_f4_xr1cnn(km, losses, **_):
    run_model_case(km, losses, P.cfg_xr1cnn(size=350), 4, "f4_xr1cnn_350.npz")
    run_model_case(km, losses, P.cfg_xr1cnn(size=310), 4, "f4_xr1cnn_310.npz")
    run_model_case(km, losses, P.cfg_xr1cnn(arch="resnet18", size=160), 2, "f4_xr1cnn_r18_160.npz")


def case_f5_mr(km, losses, **_):
    run_model_case(km, losses, P.cfg_mr1(shape=(160, 160, 64)), 1, "f5_mr1_rc_s64.npz")
    run_model_case(km, losses, P.cfg_mr1(shape=(64, 96, 32), dims_view="cs", depth=1), 2, "f5_mr1_cs.npz")
    run_model_case(km, losses, P.cfg_mr1(shape=(64, 96, 32), dims_view="rs", depth=1), 2, "f5_mr1_rs.npz")
    run_model_case(km, losses, P.cfg_mr2(), 2, "f5_mr2.npz")
    run_model_case(km, losses, P.cfg_xr1mr1(), 2, "f5_xr1mr1.npz")
    run_model_case(km, losses, P.cfg_xr1mr2(), 2, "f5_xr1mr2.npz")
    case_f5_nogap(km, losses)


def case_f5_nogap(km, losses, **_):
    """with_gap: false -- the (h, w) grid of the last stage stays as tokens (_mrN_cnn_trf.py:32-40, _xr1mrN.py:64-81)"""
    run_model_case(km, losses, P.cfg_mr1(shape=(64, 96, 32), with_gap=False, depth=1), 1, "f5_mr1_nogap.npz")
    run_model_case(km, losses, P.cfg_xr1mr1(xr=(96, 96), mr=(64, 64, 32), with_gap=False), 1, "f5_xr1mr1_nogap.npz")


def case_f6_full(km, losses, **_):
    run_model_case(km, losses, P.cfg_full(), 2, "f6_full_native_b2.npz", adam_steps=3)


def case_f7_focal(losses, **_):
    logits = np.concatenate([P.make_input("focal", (60, 2)) * 3,
                             np.array([[30, -30], [-30, 30], [0, 0], [50, 50]], dtype=np.float32)])
    tgt = P.make_target("focal_t", 64)[:, 0]
    lt = t(logits).requires_grad_(True)
    out = {"logits": logits, "target": tgt}
    for red in ("mean", "sum"):
        fl = losses.FocalLoss(reduction=red, gamma=2.0)
        lt.grad = None
        loss = fl(input=lt, target=t(tgt))
        loss.backward()
        out[f"focal_{red}:loss"] = np.float64(loss.item())
        out[f"focal_{red}:dlogits"] = lt.grad.numpy().copy()
    ce = losses.CrossEntropyLoss(num_classes=2)
    lt.grad = None
    loss = ce(lt, t(tgt))
    loss.backward()
    out["ce:loss"] = np.float64(loss.item())
    out["ce:dlogits"] = lt.grad.numpy().copy()
    # the documented input of FocalLoss.forward is (b, ch, d0, d1) with target (b, d0, d1) (_losses.py:91-94), and both
    # losses take class weights (:56-57,36): 3 classes on a 5 x 4 grid, weights (0.2, 1.0, 3.0)
    lg4 = P.make_input("focal4", (3, 3, 5, 4)) * 2
    tg4 = (np.abs(P.make_input("focal4_t", (3, 5, 4))) * 1.7).astype(np.int64) % 3
    cw = np.array([0.2, 1.0, 3.0], dtype=np.float32)
    out.update({"nd:logits": lg4, "nd:target": tg4, "nd:class_weight": cw})
    for tag, weight in (("nd", None), ("ndw", t(cw))):
        for red in ("mean", "sum"):
            l4 = t(lg4).requires_grad_(True)
            fl = losses.FocalLoss(reduction=red, gamma=2.0, num_classes=3, class_weight=weight)
            loss = fl(input=l4, target=t(tg4))
            loss.backward()
            out[f"{tag}:focal_{red}:loss"] = np.float64(loss.item())
            out[f"{tag}:focal_{red}:dlogits"] = l4.grad.numpy().copy()
        l4 = t(lg4).requires_grad_(True)
        loss = losses.CrossEntropyLoss(num_classes=3, class_weight=weight)(l4, t(tg4))
        loss.backward()
        out[f"{tag}:ce:loss"] = np.float64(loss.item())
        out[f"{tag}:ce:dlogits"] = l4.grad.numpy().copy()
    np.savez_compressed(HERE / "f7_focal.npz", **out)
    print("  wrote f7_focal.npz")


def case_f8_interp(preproc, **_):
    out = {}
    x = t(P.make_input("interp_xr", (2, 1, 70, 50)))
    out["xr"] = preproc.PTInterpolate(scale_factor=(0.5, 0.5))(x).numpy()
    v = t(P.make_input("interp_mr", (2, 1, 36, 28, 26)))
    out["mr_half"] = preproc.PTInterpolate(scale_factor=(0.5, 0.5, 0.5))(v).numpy()
    out["mr_keep"] = preproc.PTInterpolate(scale_factor=(0.5, 0.5, 1.0))(v).numpy()
    # any scale factor (the transform takes whatever the config's `downscale` holds): odd sizes, up- and down-scaling,
    # several channels and the 3-D (B, CH, D0) "linear" rank.  (The transform's MASK branch passes align_corners=False together
    # with mode="nearest", which torch refuses -- ValueError: it is dead code in the reference, recorded below as such.)
    x2 = t(P.make_input("interp_xr2", (2, 3, 37, 29)))
    v2 = t(P.make_input("interp_mr2", (1, 2, 19, 23, 11)))
    l2 = t(P.make_input("interp_lin", (2, 2, 41)))
    for tag, img, sf in (("xr_075", x2, (0.75, 0.75)), ("xr_up", x2, (1.5, 2.0)), ("xr_mix", x, (0.3, 0.85)),
                         ("mr_mix", v2, (0.6, 0.8, 1.0)), ("mr_up", v2, (1.3, 0.5, 2.0)), ("lin", l2, (0.4,))):
        out[tag] = preproc.PTInterpolate(scale_factor=sf)(img).numpy()
    try:
        preproc.PTInterpolate(scale_factor=(0.75, 0.6))(x2, (x2 > 0.3).float())
        out["mask_branch"] = np.array("returns")
    except ValueError as e:
        out["mask_branch"] = np.array("ValueError: " + str(e))
    np.savez_compressed(HERE / "f8_interp.npz", **out)
    print("  wrote f8_interp.npz")


AUG_STATES = [(0.1, 0.2, 0.3, 1.7), (0.9, 0.1, 0.2, 0.6), (0.4, -0.26, 0.8, 1.2), (0.7, 0.15, 0.9, 0.8)]   # (p_rot, theta, p_gamma, gamma)


def case_f12_augment(preproc, **_):
    """the reference's per-sample train pipeline of one modality (_data_provider.py:295-335) with pinned random states"""
    out = {"states": np.asarray(AUG_STATES, dtype=np.float64)}
    for tag, shape, rot_cls, mean, std in (("mr", (4, 1, 24, 20, 6), preproc.PTRotate3DInSlice, 0.257, 0.235),
                                           ("xr", (4, 1, 28, 22), preproc.PTRotate2D, 0.543, 0.296)):
        raw = np.abs(P.make_input("aug_" + tag, shape)) * 300.0 + 5.0
        res = []
        for b, (p_rot, theta, p_gam, gamma) in enumerate(AUG_STATES):
            rot = rot_cls(degree_range=[-15., 15.], prob=0.5)
            rot.state = {"p": p_rot, "theta": torch.tensor(theta)}
            gam = preproc.PTGammaCorrection(gamma_range=(0.5, 2.0), prob=0.5, clip_to_unit=False)
            gam.state = {"p": p_gam, "gamma": gamma}
            x = t(raw[b].astype(np.float32))
            for tf in (preproc.PTToUnitRange(), rot, gam, preproc.PTNormalize(mean=[mean, ], std=[std, ])):
                x = tf(x)
            res.append(x.numpy())
        out[tag] = np.stack(res)
        out[tag + ":norm"] = np.array([mean, std])
    np.savez_compressed(HERE / "f12_augment.npz", **out)
    print("  wrote f12_augment.npz")


def case_f13_modal_abl(km, **_):
    """Explain regime (eval_prog_fus.py:410-479).  captum is absent here, so its FeatureAblation rule for the
    reference's call (one feature id per input, zero baselines, one perturbation per evaluation) is applied by hand
    to the REFERENCE model's own forwards: attr[b, m] = f(x)[b, y_b] - f(x, modality m zeroed)[b, y_b]."""
    t0 = time.time()
    cfg = P.cfg_full(xr=(160, 160), mr1=(96, 96, 6), mr2=(96, 96, 5), depth=1)
    cfg["output_type"] = "main"                      # the tensor-returning mode the reference keeps for captum
    B, seed = 3, 77
    torch.manual_seed(0)
    model = km.dict_models[cfg["name"]](config=Cfg(cfg), path_weights=None)
    P.fill_state_dict(model.state_dict())
    model.eval()
    xs = [t(a) for a in P.model_inputs(cfg, B, seed)]
    y = t(P.make_target("target", B, seed))
    logits = []
    with torch.no_grad():
        logits.append(model(*xs).reshape(B, -1))
        for m in range(len(xs)):
            logits.append(model(*[torch.zeros_like(x) if j == m else x for j, x in enumerate(xs)]).reshape(B, -1))
    sel = [lg.gather(1, y.long()) for lg in logits]
    attrs = torch.cat([sel[0] - s for s in sel[1:]], dim=1)
    pc = attrs / torch.sum(torch.abs(attrs), dim=1, keepdim=True)
    percent = np.round(np.abs(pc.numpy()) * 100., decimals=3)
    np.savez_compressed(HERE / "f13_modal_abl.npz", B=np.int64(B), seed=np.int64(seed),
                        cfg_json=np.array(json.dumps(cfg)), torch_version=np.array(torch.__version__),
                        logits=torch.stack(logits).numpy(), target=y.numpy(), attrs=attrs.numpy(), percent=percent)
    print(f"  wrote f13_modal_abl.npz in {time.time() - t0:.1f}s; attrs=\n{attrs.numpy()}\npercent=\n{percent}")


def case_f9_sched(optims, **_):
    p = [torch.nn.Parameter(torch.zeros(1))]
    tab = {}
    opt = torch.optim.Adam(p, lr=1e-4)
    s = optims.dict_schedulers["CustomWarmupStaticDecayLR"](optimizer=opt, epochs_warmup=5, epochs_static=100,
                                                            epochs_decay=1)
    lrs = []
    for e in range(121):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        s.step()
    tab["static_decay"] = [repr(float(v)) for v in lrs]
    opt = torch.optim.Adam(p, lr=1e-3)
    s = optims.dict_schedulers["CustomWarmupMultiStepLR"](optimizer=opt, epochs_warmup=5, mstep_milestones=[20, 40])
    lrs = []
    for e in range(121):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        s.step()
    tab["multistep"] = [repr(float(v)) for v in lrs]
    tab["optimizer_keys"] = sorted(optims.dict_optimizers)
    tab["scheduler_keys"] = sorted(optims.dict_schedulers)
    (HERE / "f9_schedules.json").write_text(json.dumps(tab, indent=0))
    print("  wrote f9_schedules.json")


def case_f11_bookkeeping(km, losses, **_):
    book = {"dict_models": sorted(km.dict_models), "dict_losses": sorted(losses.dict_losses)}
    cfgs = {"XR1Cnn": P.cfg_xr1cnn(), "MR1CnnTrf": P.cfg_mr1(), "MR2CnnTrf": P.cfg_mr2((160, 160, 64), (160, 160, 32), 4),
            "XR1MR1CnnTrf": P.cfg_xr1mr1((350, 350), (160, 160, 64), 4),
            "XR1MR2CnnTrf": P.cfg_xr1mr2((350, 350), (160, 160, 64), (160, 160, 32), 4), "XR1MR2C1CnnTrf": P.cfg_full()}
    for name, cfg in cfgs.items():
        torch.manual_seed(0)
        m = km.dict_models[name](config=Cfg(cfg), path_weights=None)
        sd = m.state_dict()
        book[name] = {
            "vs": {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in getattr(m, "vs", {}).items()},
            "state_dict": [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()],
            "num_params": int(sum(p.numel() for p in m.parameters())),
        }
    # non-default config variants whose bookkeeping differs (with_gap false)
    cfg = P.cfg_mr1(shape=(160, 160, 64), with_gap=False)
    m = km.dict_models["MR1CnnTrf"](config=Cfg(cfg), path_weights=None)
    book["MR1CnnTrf_nogap"] = {"vs": {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in m.vs.items()}}
    (HERE / "f11_bookkeeping.json").write_text(json.dumps(book))
    print("  wrote f11_bookkeeping.json")


def case_f14_fullsize(km, tv, losses, **_):
    """BASELINE.json's full sizes (384^2 slices, 160 slices per volume, 483 fusion tokens): values, not only properties"""
    t0 = time.time()
    out = {}
    arch, shape = "resnet50", (2, 1, 384, 384)

    def build():
        torch.manual_seed(0)
        net = getattr(tv, arch)(pretrained=False)
        tr = torch.nn.Sequential(*list(net.children())[:-1])
        P.fill_state_dict(tr.state_dict())
        return tr
    trunk = build()
    x = t(P.make_input("trunk", shape)).repeat(1, 3, 1, 1)
    tag = f"{arch}_{shape[0]}x{shape[2]}x{shape[3]}"
    trunk.eval()
    with torch.no_grad():
        out[tag + ":eval"] = trunk(x).numpy()

    def run(m, dt):
        m.train()
        yy = m(x.to(dt))
        (yy * t(P.make_input("trunkg", tuple(yy.shape))).to(dt)).sum().backward()
        return yy.detach(), {k: p.grad.detach() for k, p in m.named_parameters()}
    y, g32 = run(trunk, torch.float32)
    out[tag + ":train"] = y.numpy()
    out.update({tag + ":" + k: v for k, v in P.summarize_tensors({"grad:" + k: v.numpy() for k, v in g32.items()}).items()})
    out.update({tag + ":" + k: v for k, v in P.summarize_tensors({"buf:" + k: b.numpy() for k, b in trunk.named_buffers()}).items()})
    out.update({tag + ":" + k: v for k, v in fp64_twin(build, run, g32, y).items()})
    np.savez_compressed(HERE / "f14_trunk384.npz", **out)
    print(f"  wrote f14_trunk384.npz in {time.time() - t0:.1f}s")
    # the fusion transformer at the synthetic-shape token count: 482 tokens + cls
    t0 = time.time()
    torch.manual_seed(0)
    f = km.FeaT(num_patches=482, patch_dim=2048, emb_dim=2048, depth=4, heads=8, mlp_dim=2048, num_classes=2, with_cls=True)
    P.fill_state_dict(f.state_dict())
    f.train()                                       # (all dropout probabilities are 0)
    xf = t(P.make_input("feat483", (2, 482, 2048))).requires_grad_(True)
    o, st, att = f(xf)
    ((o * t(P.make_input("feat483go", tuple(o.shape)))).sum() + (st * t(P.make_input("feat483gs", tuple(st.shape)))).sum() * 1e-2).backward()
    fo = {"outputs": o.detach().numpy()}
    fo.update(P.summarize_tensors({"states": st.detach().numpy(), "attn0": att[0].detach().numpy(),
                                   "attn3": att[3].detach().numpy(), "dx": xf.grad.numpy()}, k=64))
    fo.update(P.summarize_tensors({"grad:" + k: p.grad.numpy() for k, p in f.named_parameters() if p.grad is not None}))
    np.savez_compressed(HERE / "f14_feat483.npz", **fo)
    print(f"  wrote f14_feat483.npz in {time.time() - t0:.1f}s")
    del f, trunk
    # BASELINE config 3's pinned class on its full tensor: one DESS volume of 160 slices x 384 x 384
    run_model_case(km, losses, P.cfg_mr1(shape=(320, 320, 160)), 1, "f14_mr1_s160_384.npz", tensor_shapes=[[384, 384, 160]], twin=False)


CASES = {
    "f1": case_f1_attention_feat, "f2": case_f2_bottleneck, "f2b": case_f2b_stage, "f3": case_f3_trunk, "f4": case_f4_xr1cnn,
    "f5": case_f5_mr, "f5g": case_f5_nogap, "f6": case_f6_full, "f7": case_f7_focal, "f8": case_f8_interp, "f9": case_f9_sched,
    "f11": case_f11_bookkeeping, "f12": case_f12_augment, "f13": case_f13_modal_abl, "f14": case_f14_fullsize,
}


def main():
    torch.set_num_threads(8)
    km, tv, preproc, losses, optims = import_reference()
    which = sys.argv[1:] or list(CASES)
    for c in which:
        print(f"[{c}]")
        CASES[c](km=km, tv=tv, preproc=preproc, losses=losses, optims=optims)


if __name__ == "__main__":
    main()
