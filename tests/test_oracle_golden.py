"""CPU: the oracle (oracle/koafusion_cpu.py) against the golden vectors produced by the imported
reference (tests/golden/make_golden.py).  This is what PINS the oracle.  Tolerances: both sides are torch
fp32 CPU, identical weights/inputs, so agreement is at accumulation-order level (1e-5 relative; gradients
through 50 train-mode BatchNorm layers 2e-4)."""
import json
import os

import numpy as np
import pytest
import torch

import procedural as P
from common import GOLDEN, cfg_of, check_summary, load, rel
from oracle import koafusion_cpu as O


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def run_oracle_case(gold, adam_steps=0, cfg_extra=None):
    cfg = cfg_of(gold)
    if cfg_extra:
        cfg = dict(cfg, **cfg_extra)
    B = int(gold["B"])
    seed = int(gold["seed"])
    m = O.OracleModel(cfg, fill=P.fill_value)
    xs = [t(a) for a in P.model_inputs(cfg, B, seed)]
    y = t(P.make_target("target", B, seed))
    out = {}
    with torch.no_grad():
        out["eval_logits"] = m(*xs, train=False).numpy()
    logits, loss = m.train_step(xs, y, optimize=False)
    out["train_logits"], out["train_loss"] = logits.numpy(), float(loss)
    named, none = {}, []
    for k, p in m.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            named["grad:" + k] = p.grad.numpy()
    out.update(P.summarize_tensors(named))
    out["none_grad_keys"] = none
    out.update(P.summarize_tensors({"buf:" + k: b.numpy() for k, b in m.named_buffers()}))
    if adam_steps:
        ps = [p for _, p in m.named_parameters()]
        losses = [float(loss)]
        with torch.no_grad():
            O.adam_step(ps, [p.grad for p in ps], m.opt_state)
        for s in range(1, adam_steps):
            _, l = m.train_step(xs, y, optimize=True)
            losses.append(float(l))
        out["adam_losses"] = losses
        out.update({"adam:" + k: v for k, v in P.summarize_tensors(
            {"param:" + k: p.detach().numpy() for k, p in m.named_parameters()}).items()})
    return out


def check_fp64_twin(gold):
    """the oracle in float64 against the reference in float64: validates the ground-truth generator the
    GPU parity tests use"""
    cfg, B, seed = cfg_of(gold), int(gold["B"]), int(gold["seed"])
    m = O.OracleModel(cfg, fill=P.fill_value, dtype=torch.float64)
    xs = [t(a) for a in P.model_inputs(cfg, B, seed)]
    y = t(P.make_target("target", B, seed))
    logits, _ = m.train_step(xs, y, optimize=False)
    assert rel(logits.numpy(), gold["train_logits64"]) < 1e-10
    got = P.summarize_tensors({"g64:" + k: p.grad.numpy() for k, p in m.named_parameters() if p.grad is not None})
    check_summary(got, gold, "g64:", 1e-8, "fp64 gradients")


def compare_case(out, gold, gtol=2e-4):
    assert rel(out["eval_logits"], gold["eval_logits"]) < 1e-5
    assert rel(out["train_logits"], gold["train_logits"]) < 1e-5
    assert abs(out["train_loss"] - float(gold["train_loss"])) < 1e-5 * max(1.0, abs(float(gold["train_loss"])))
    assert sorted(out["none_grad_keys"]) == sorted(str(k) for k in gold["none_grad_keys"])
    check_summary(out, gold, "grad:", gtol, "param grads")
    check_summary(out, gold, "buf:", 1e-5, "BN buffers")


SMALL = ["f4_xr1cnn_350.npz", "f4_xr1cnn_310.npz", "f4_xr1cnn_r18_160.npz", "f5_mr1_cs.npz", "f5_mr1_rs.npz",
         "f5_mr2.npz", "f5_xr1mr1.npz", "f5_xr1mr2.npz", "f5_mr1_rc_s64.npz", "f5_mr1_nogap.npz",
         "f5_xr1mr1_nogap.npz"]


@pytest.mark.parametrize("fname", SMALL)
def test_oracle_models(fname):
    gold = load(fname)
    compare_case(run_oracle_case(gold), gold)
    if fname in ("f4_xr1cnn_r18_160.npz", "f5_mr1_cs.npz", "f5_xr1mr2.npz"):
        check_fp64_twin(gold)


@pytest.mark.skipif(not (GOLDEN / "f6_full_native_b2.npz").exists(), reason="fixture missing")
def test_oracle_full_fusion():
    gold = load("f6_full_native_b2.npz")
    steps = 3 if os.environ.get("KOAF_SLOW") else 0
    out = run_oracle_case(gold, adam_steps=steps)
    compare_case(out, gold)
    if steps:
        assert np.allclose(out["adam_losses"], gold["adam_losses"], rtol=1e-4)
        check_summary(out, gold, "adam:", 1e-5, "params after 3 Adam steps")


@pytest.mark.skipif(not (GOLDEN / "f6_full_native_b2.npz").exists(), reason="fixture missing")
def test_oracle_generic_hierarchy_reproduces_the_reference_class():
    """The generic hierarchical statement the headline model (XR1MR3C1CnnTrf) runs, at (n_xr, n_mr) = (1, 2) under the
    reference's own XR1MR2C1CnnTrf config, against fixture F6 from the imported reference
    (koafusion/models/_xrNmrMcP.py:33-264): same state-dict keys and shapes, same `vs` bookkeeping, eval / train logits,
    loss, gradient-less parameter set, gradients and BatchNorm buffers.  This pins the COMPOSITION (token order, per-MRI
    aggregators without cls token, `_agg_final` sizing); three MRI are one more iteration of the same loop."""
    gold = load("f6_full_native_b2.npz")
    cfg = cfg_of(gold)
    spec_ref, vs_ref = O.model_spec(cfg)
    spec_gen, vs_gen = O.model_spec(dict(cfg, ext_pattern=(1, 2)))
    assert [(k, tuple(s), d) for k, s, d in spec_gen] == [(k, tuple(s), d) for k, s, d in spec_ref]
    # (`vs` names: the reference writes ONE `fe12_out_ch` for its two MRI trunks, the generic class one per trunk)
    assert {k: v for k, v in vs_gen.items() if k in vs_ref} == {k: v for k, v in vs_ref.items() if k in vs_gen}
    assert set(vs_ref) - set(vs_gen) == {"fe12_out_ch"} and vs_gen["fe1_out_ch"] == vs_gen["fe2_out_ch"] == vs_ref["fe12_out_ch"]
    compare_case(run_oracle_case(gold, cfg_extra=dict(ext_pattern=(1, 2))), gold)


def test_oracle_layer1_f2b():
    """fixture F2b (ResNet-50 layer1 of the imported reference, train-mode forward + backward): the oracle's three Bottlenecks
    element-wise on the ReLU-safe small case (1e-5 of each tensor's largest magnitude: both sides are torch CPU fp32)"""
    g = load("f2b_layer1.npz")
    N, C, H, W = (int(v) for v in g["small:shape"])
    seed = int(g["small:seed"])
    spec = [k for k in g.files if k.startswith("small:grad:") or k.startswith("small:buf:")]
    sd = {}
    for k in spec:
        name = k.split(":", 2)[2]
        v = t(P.fill_value(name, g[k].shape, name.endswith("num_batches_tracked")))
        sd["l." + name] = v.reshape(g[k].shape).clone()
    for k in sd:
        if O.is_param(k):
            sd[k].requires_grad_(True)
    x = torch.relu(t(P.make_input("f2bx_small", (N, C, H, W), seed=seed))).requires_grad_(True)
    y = x + 0
    for b in range(3):
        y = O._bottleneck(y, sd, f"l.{b}", True, 1, 1)
    (y * t(P.make_input("f2bg_small", tuple(y.shape), seed=seed))).sum().backward()

    def mx(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / np.abs(b).max())
    assert mx(y.detach().numpy(), g["small:train"]) < 1e-5 and mx(x.grad.numpy(), g["small:dx"]) < 1e-5
    for k in spec:
        name = k.split(":", 2)[2]
        if k.startswith("small:grad:"):
            assert mx(sd["l." + name].grad.numpy(), g[k]) < 1e-5, name
        elif not name.endswith("num_batches_tracked"):
            assert mx(sd["l." + name].detach().numpy(), g[k]) < 1e-5, name


def test_oracle_attention_feat():
    g = load("f1_attention_feat.npz")
    for dim, heads, n in ((64, 4, 25), (2048, 8, 12)):
        sd = {"a.to_qkv.weight": t(P.fill_value("to_qkv.weight", (3 * dim, dim))),
              "a.to_out.0.weight": t(P.fill_value("to_out.0.weight", (dim, dim))),
              "a.to_out.0.bias": t(P.fill_value("to_out.0.bias", (dim,)))}
        x = t(P.make_input(f"att{dim}", (2, n, dim))).requires_grad_(True)
        o, a = O.attention(x, sd, "a", heads)
        (o * t(P.make_input(f"attg{dim}", (2, n, dim)))).sum().backward()
        assert rel(o.detach().numpy(), g[f"att{dim}:out"]) < 1e-5
        assert rel(a.detach().numpy(), g[f"att{dim}:attn"]) < 1e-5
        assert rel(x.grad.numpy(), g[f"att{dim}:dx"]) < 1e-5
    for with_cls in (True, False):
        spec = O.feat_spec("f", 25, 64, 2, 128, 2, with_cls)
        sd = {k: t(P.fill_value(k[2:], s, dt == torch.int64)) for k, s, dt in spec}
        x = t(P.make_input("feat", (3, 25, 64)))
        o, s, a = O.feat(x, sd, "f", 2, 4, with_cls)
        tag = f"feat_cls{int(with_cls)}"
        assert rel(o.numpy(), g[tag + ":outputs"]) < 1e-5
        assert rel(s.numpy(), g[tag + ":states"]) < 1e-5
        assert rel(a[0].numpy(), g[tag + ":attn0"]) < 1e-5


F2_CASES = (("s1", 256, 64, 1, 1, 64), ("s2ds", 256, 128, 2, 1, 64), ("g32", 256, 64, 1, 32, 4), ("g32s2ds", 256, 128, 2, 32, 4))


def f2_state(tag, inpl, planes, stride, groups, bw):
    """state dict of one reference Bottleneck (_torchvision.py:101-115), filled by key name like the fixture's"""
    width = int(planes * (bw / 64.0)) * groups
    shapes = {"conv1.weight": (width, inpl, 1, 1), "conv2.weight": (width, width // groups, 3, 3),
              "conv3.weight": (planes * 4, width, 1, 1)}
    bns = {"bn1": width, "bn2": width, "bn3": planes * 4}
    if stride != 1 or inpl != planes * 4:
        shapes["downsample.0.weight"] = (planes * 4, inpl, 1, 1)
        bns["downsample.1"] = planes * 4
    sd = {"b." + k: t(P.fill_value(k, s)) for k, s in shapes.items()}
    for bn, c in bns.items():
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            sd[f"b.{bn}.{leaf}"] = t(P.fill_value(f"{bn}.{leaf}", (c,)))
        sd[f"b.{bn}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return sd


def test_oracle_bottleneck_f2():
    """F2: one Bottleneck, full tensors, element-wise (max-norm relative to the tensor's largest magnitude) at 1e-5"""
    g = load("f2_bottleneck.npz")

    def mx(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.abs(a - b).max() / np.abs(b).max())
    for tag, inpl, planes, stride, groups, bw in F2_CASES:
        seed = int(g[tag + ":seed"])
        x = torch.relu(t(P.make_input("f2x_" + tag, (2, inpl, 12, 12), seed=seed)))
        sd = f2_state(tag, inpl, planes, stride, groups, bw)
        with torch.no_grad():
            ye = O._bottleneck(x, {k: v.clone() for k, v in sd.items()}, "b", False, stride, groups)
        assert mx(ye.numpy(), g[tag + ":eval"]) < 1e-5
        for k in sd:
            if O.is_param(k):
                sd[k].requires_grad_(True)
        xi = x.clone().requires_grad_(True)
        y = O._bottleneck(xi, sd, "b", True, stride, groups)
        (y * t(P.make_input("f2g_" + tag, tuple(y.shape), seed=seed))).sum().backward()
        assert mx(y.detach().numpy(), g[tag + ":train"]) < 1e-5, tag
        assert mx(xi.grad.numpy(), g[tag + ":dx"]) < 1e-5, tag
        for k, v in sd.items():
            if O.is_param(k):
                assert mx(v.grad.numpy(), g[f"{tag}:grad:{k[2:]}"]) < 1e-5, (tag, k)
            else:
                assert mx(v.detach().numpy(), g[f"{tag}:buf:{k[2:]}"]) < 1e-5 or k.endswith("num_batches_tracked"), (tag, k)
        assert int(sd["b.bn1.num_batches_tracked"]) == int(g[tag + ":buf:bn1.num_batches_tracked"]) == 1


def test_oracle_trunks():
    g = load("f3_trunk.npz")
    for arch, shape in (("resnet50", (4, 1, 160, 160)), ("resnet50", (2, 1, 96, 112)),
                        ("resnext50_32x4d", (2, 1, 130, 130)), ("resnet18", (2, 1, 96, 96)),
                        ("resnet34", (2, 1, 64, 96))):
        tag = f"{arch}_{shape[0]}x{shape[2]}x{shape[3]}"
        spec = O.trunk_spec("t", arch)
        sd = {k: t(P.fill_value(k[2:], s, dt == torch.int64)).reshape(s) for k, s, dt in spec}
        x = t(P.make_input("trunk", shape))
        with torch.no_grad():
            ye = O.trunk(x, sd, "t", arch, False)
        assert rel(ye.numpy(), g[tag + ":eval"]) < 1e-5
        for k in sd:
            if O.is_param(k):
                sd[k].requires_grad_(True)
        y = O.trunk(x, sd, "t", arch, True)
        (y * t(P.make_input("trunkg", tuple(y.shape)))).sum().backward()
        assert rel(y.detach().numpy(), g[tag + ":train"]) < 1e-5
        got = P.summarize_tensors({"grad:" + k[2:]: v.grad.numpy() for k, v in sd.items() if O.is_param(k)})
        got.update(P.summarize_tensors({"buf:" + k[2:]: v.numpy() for k, v in sd.items() if not O.is_param(k)}))
        got = {tag + ":" + k: v for k, v in got.items()}
        check_summary(got, g, tag + ":grad:", 2e-4, tag)
        check_summary(got, g, tag + ":buf:", 1e-5, tag)


def test_oracle_fullsize_fixtures():
    """the oracle against the full-size fixtures F14 it can afford on the CPU suite: the ResNet-50 trunk forward on 2 x 384 x 384
    (eval and train outputs) and FeaT at 482 + 1 tokens (outputs, states, attention, every gradient); the 160-slice MR1CnnTrf
    fixture takes minutes and 45 GB and is checked on the GPU side only (tests/test_fullsize_values_gpu.py)"""
    g = load("f14_trunk384.npz")
    arch, shape = "resnet50", (2, 1, 384, 384)
    tag = f"{arch}_{shape[0]}x{shape[2]}x{shape[3]}"
    spec = O.trunk_spec("t", arch)
    sd = {k: t(P.fill_value(k[2:], s, dt == torch.int64)).reshape(s) for k, s, dt in spec}
    x = t(P.make_input("trunk", shape))
    with torch.no_grad():
        assert rel(O.trunk(x, sd, "t", arch, False).numpy(), g[tag + ":eval"]) < 1e-5
        assert rel(O.trunk(x, {k: v.clone() for k, v in sd.items()}, "t", arch, True).numpy(), g[tag + ":train"]) < 1e-5
    g = load("f14_feat483.npz")
    spec = O.feat_spec("f", 482, 2048, 4, 2048, 2, True)
    sd = {k: t(P.fill_value(k[2:], s, dt == torch.int64)) for k, s, dt in spec}
    for k in sd:
        if O.is_param(k):
            sd[k].requires_grad_(True)
    xf = t(P.make_input("feat483", (2, 482, 2048))).requires_grad_(True)
    o, st, att = O.feat(xf, sd, "f", 4, 8, True)
    ((o * t(P.make_input("feat483go", tuple(o.shape)))).sum() + (st * t(P.make_input("feat483gs", tuple(st.shape)))).sum() * 1e-2).backward()
    assert rel(o.detach().numpy(), g["outputs"]) < 1e-5
    got = P.summarize_tensors({"states": st.detach().numpy(), "attn0": att[0].detach().numpy(),
                               "attn3": att[3].detach().numpy(), "dx": xf.grad.numpy()}, k=64)
    got.update(P.summarize_tensors({"grad:" + k[2:]: v.grad.numpy() for k, v in sd.items() if O.is_param(k) and v.grad is not None}))
    for key in ("states:", "attn0:", "attn3:", "dx:", "grad:"):
        check_summary(got, g, key, 1e-4 if key == "grad:" else 1e-5, "FeaT n=483 " + key)


def test_oracle_focal_interp_sched():
    g = load("f7_focal.npz")
    lt = t(g["logits"]).requires_grad_(True)
    for red in ("mean", "sum"):
        lt.grad = None
        loss = O.focal_loss(lt, t(g["target"]), 2.0, red)
        loss.backward()
        assert abs(loss.item() - float(g[f"focal_{red}:loss"])) < 1e-6 * max(1, abs(float(g[f"focal_{red}:loss"])))
        assert rel(lt.grad.numpy(), g[f"focal_{red}:dlogits"]) < 1e-6
    # (b, ch, d0, d1) logits and class weights (_losses.py:56-57,91-94), both losses
    lg4, tg4, cw = t(g["nd:logits"]), t(g["nd:target"]), t(g["nd:class_weight"])
    for tag, w in (("nd", None), ("ndw", cw)):
        for red in ("mean", "sum"):
            l4 = lg4.clone().requires_grad_(True)
            loss = O.focal_loss(l4, tg4, 2.0, red, class_weight=w)
            loss.backward()
            assert abs(loss.item() - float(g[f"{tag}:focal_{red}:loss"])) < 1e-6 * max(1, abs(float(g[f"{tag}:focal_{red}:loss"])))
            assert rel(l4.grad.numpy(), g[f"{tag}:focal_{red}:dlogits"]) < 1e-6
        l4 = lg4.clone().requires_grad_(True)
        loss = O.ce_loss(l4, tg4, class_weight=w)
        loss.backward()
        assert abs(loss.item() - float(g[f"{tag}:ce:loss"])) < 1e-6 and rel(l4.grad.numpy(), g[f"{tag}:ce:dlogits"]) < 1e-6
    g = load("f8_interp.npz")
    x2, v2, l2 = t(P.make_input("interp_xr2", (2, 3, 37, 29))), t(P.make_input("interp_mr2", (1, 2, 19, 23, 11))), t(P.make_input("interp_lin", (2, 2, 41)))
    for tag, img, sf in (("xr_075", x2, (0.75, 0.75)), ("xr_up", x2, (1.5, 2.0)), ("xr_mix", t(P.make_input("interp_xr", (2, 1, 70, 50))), (0.3, 0.85)),
                         ("mr_mix", v2, (0.6, 0.8, 1.0)), ("mr_up", v2, (1.3, 0.5, 2.0)), ("lin", l2, (0.4,))):
        got = O.interpolate(img, sf).numpy()
        assert got.shape == g[tag].shape and rel(got, g[tag]) < 1e-7, tag
    assert str(g["mask_branch"]).startswith("ValueError")        # the reference's mask branch is dead code
    assert rel(O.interpolate(t(P.make_input("interp_xr", (2, 1, 70, 50))), (0.5, 0.5)).numpy(), g["xr"]) < 1e-7
    v = t(P.make_input("interp_mr", (2, 1, 36, 28, 26)))
    assert rel(O.interpolate(v, (0.5, 0.5, 0.5)).numpy(), g["mr_half"]) < 1e-7
    assert rel(O.interpolate(v, (0.5, 0.5, 1.0)).numpy(), g["mr_keep"]) < 1e-7
    tab = json.loads((GOLDEN / "f9_schedules.json").read_text())
    # bit-exact schedule bookkeeping: lr = base_lr * lambda(epoch), compared through repr()
    assert [repr(1e-4 * O.lr_factor_static_decay(e, 5, 100)) for e in range(121)] == tab["static_decay"]
    assert [repr(1e-3 * O.lr_factor_multistep(e, 5, [20, 40])) for e in range(121)] == tab["multistep"]


def test_oracle_bookkeeping():
    book = json.loads((GOLDEN / "f11_bookkeeping.json").read_text())
    cfgs = {"XR1Cnn": P.cfg_xr1cnn(), "MR1CnnTrf": P.cfg_mr1(),
            "MR2CnnTrf": P.cfg_mr2((160, 160, 64), (160, 160, 32), 4),
            "XR1MR1CnnTrf": P.cfg_xr1mr1((350, 350), (160, 160, 64), 4),
            "XR1MR2CnnTrf": P.cfg_xr1mr2((350, 350), (160, 160, 64), (160, 160, 32), 4),
            "XR1MR2C1CnnTrf": P.cfg_full()}
    for name, cfg in cfgs.items():
        spec, vs = O.model_spec(cfg)
        ref = book[name]
        assert [[k, list(s), str(dt)] for k, s, dt in spec] == ref["state_dict"], name
        assert {k: (list(v) if isinstance(v, tuple) else v) for k, v in vs.items()} == ref["vs"], name
    _, vs = O.model_spec(P.cfg_mr1(shape=(160, 160, 64), with_gap=False))
    assert {k: (list(v) if isinstance(v, tuple) else v) for k, v in vs.items()} == book["MR1CnnTrf_nogap"]["vs"]


def test_f12_augment_pipeline():
    """oracle.augment_sample == the reference's own transform classes with pinned random states (F12)"""
    g = load("f12_augment.npz")
    for tag, shape in (("mr", (4, 1, 24, 20, 6)), ("xr", (4, 1, 28, 22))):
        raw = np.abs(P.make_input("aug_" + tag, shape)) * 300.0 + 5.0
        mean, std = g[tag + ":norm"]
        for b, st in enumerate(g["states"]):
            got = O.augment_sample(torch.from_numpy(raw[b].astype(np.float32)), tuple(st), float(mean), float(std))
            assert got.shape == g[tag][b].shape
            assert np.abs(got.numpy() - g[tag][b]).max() < 1e-6, (tag, b)
