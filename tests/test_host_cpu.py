"""CPU: host logic of the product and the C-ABI surface (no compute calls: there is no GPU here).
 * libkoaf.so loads and exports every symbol include/koaf.h declares; ctypes struct layout == the C layout
 * registries have the reference's keys; bookkeeping (`vs` dicts, state-dict keys/shapes/dtypes, parameter
   counts) of all six models is bit-exact against fixture F11 taken from the imported reference
 * LR schedules are bit-exact (repr) against fixture F9
 * the product fails loudly on CPU tensors (no silent fallback) and never imports the oracle
"""
import ctypes
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch

import procedural as P
from common import GOLDEN

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from oaprogressionmmf_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 55
    handle = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [n for n in protos if not hasattr(handle, n)]
    assert not missing, missing
    L = _lib.lib()
    assert L.koaf_version() >= 100
    # an argument error comes back as a status + message, not a crash
    assert L.koaf_slab_reduce(None, 0, 0, None, None) != 0
    assert b"koaf_slab_reduce" in L.koaf_last_error()


def test_struct_layout_matches_c(tmp_path):
    from oaprogressionmmf_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include "koaf.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(KoafGemm),sizeof(KoafOperand),offsetof(KoafGemm,C),offsetof(KoafGemm,stats),'
                   'offsetof(KoafGemm,cmap),offsetof(KoafOperand,sc));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    c = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    py = [ctypes.sizeof(_lib.KoafGemm), ctypes.sizeof(_lib.KoafOperand), _lib.KoafGemm.C.offset,
          _lib.KoafGemm.stats.offset, _lib.KoafGemm.cmap.offset, _lib.KoafOperand.sc.offset]
    assert c == py


def _cfgs():
    return {"XR1Cnn": P.cfg_xr1cnn(), "MR1CnnTrf": P.cfg_mr1(),
            "MR2CnnTrf": P.cfg_mr2((160, 160, 64), (160, 160, 32), 4),
            "XR1MR1CnnTrf": P.cfg_xr1mr1((350, 350), (160, 160, 64), 4),
            "XR1MR2CnnTrf": P.cfg_xr1mr2((350, 350), (160, 160, 64), (160, 160, 32), 4),
            "XR1MR2C1CnnTrf": P.cfg_full()}


def test_registries_and_bookkeeping_match_reference():
    from oaprogressionmmf_amd.config import ConfigDict
    from oaprogressionmmf_amd.models import REFERENCE_MODELS, dict_fes, dict_models
    from oaprogressionmmf_amd.various import dict_losses, dict_optimizers, dict_schedulers
    book = json.loads((GOLDEN / "f11_bookkeeping.json").read_text())
    sched = json.loads((GOLDEN / "f9_schedules.json").read_text())
    assert sorted(REFERENCE_MODELS) == book["dict_models"]
    assert sorted(set(dict_models) - set(REFERENCE_MODELS)) == ["MR1C1CnnTrf", "XR1C1Cnn", "XR1MR3C1CnnTrf"]
    assert sorted(dict_losses) == book["dict_losses"]
    assert sorted(dict_optimizers) == sched["optimizer_keys"]
    assert sorted(dict_schedulers) == sched["scheduler_keys"]
    assert sorted(dict_fes) == sorted(["squeezenet1_0", "vgg16", "densenet161", "inception_v3", "resnet18",
                                       "resnet34", "resnet50", "resnext50_32x4d"])
    for name, cfg in _cfgs().items():
        m = dict_models[name](config=ConfigDict(cfg), path_weights=None)
        ref = book[name]
        assert [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()] == ref["state_dict"], name
        assert sum(p.numel() for p in m.parameters()) == ref["num_params"], name
        vs = {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in getattr(m, "vs", {}).items()}
        assert vs == ref["vs"], name
    m = dict_models["MR1CnnTrf"](config=ConfigDict(P.cfg_mr1(shape=(160, 160, 64), with_gap=False)), path_weights=None)
    assert {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in m.vs.items()} == book["MR1CnnTrf_nogap"]["vs"]


def test_extension_models_bookkeeping_matches_oracle_statement():
    """the three registry extensions (no reference class): state-dict keys/shapes/dtypes and `vs` equal the
    oracle's statement of the same definitions; the 2-MRI slots of the 3-MRI model are key-compatible with the
    reference-named XR1MR2C1CnnTrf"""
    from oracle import koafusion_cpu as O
    from oaprogressionmmf_amd.config import ConfigDict
    from oaprogressionmmf_amd.models import dict_models
    cfgs = [P.cfg_xr1c1(), P.cfg_mr1c1(), P.cfg_xr1mr3c1(depth=1)]
    for cfg in cfgs:
        m = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None)
        spec, vs = O.model_spec(cfg)
        got = [(k, tuple(v.shape), v.dtype) for k, v in m.state_dict().items()]
        assert sorted(got) == sorted((k, tuple(s), d) for k, s, d in spec), cfg["name"]
        norm = lambda d: {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in d.items()}  # noqa: E731
        assert norm(m.vs) == norm(vs), cfg["name"]
    m3 = dict_models["XR1MR3C1CnnTrf"](config=ConfigDict(P.cfg_xr1mr3c1(mr2=(160, 160, 25), depth=1)), path_weights=None)
    m2 = dict_models["XR1MR2C1CnnTrf"](config=ConfigDict(P.cfg_full(depth=1)), path_weights=None)
    k3, k2 = dict(m3.state_dict()), dict(m2.state_dict())
    for k, v in k2.items():
        if k.startswith(("_fe0.", "_fe1.", "_fe2.", "_agg_1.", "_agg_2.")):
            assert k in k3 and k3[k].shape == v.shape, k
    with pytest.raises(ValueError):
        dict_models["XR1MR3C1CnnTrf"](config=ConfigDict(P.cfg_full()), path_weights=None)   # 4 inputs given


def test_config_errors_like_reference():
    from oaprogressionmmf_amd.config import ConfigDict
    from oaprogressionmmf_amd.models import dict_models
    bad = P.cfg_full(xr=(310, 310))          # 310 is not in the reference's size table (_xrNmrMcP.py:104-106)
    with pytest.raises(AssertionError):
        dict_models["XR1MR2C1CnnTrf"](config=ConfigDict(bad), path_weights=None)
    with pytest.raises(ValueError):
        dict_models["MR1CnnTrf"](config=ConfigDict(P.cfg_mr1(dims_view="xy")), path_weights=None)
    with pytest.raises(ValueError):
        dict_models["MR1CnnTrf"](config=ConfigDict(P.cfg_mr1(arch="resnext50_32x4d")), path_weights=None)
    with pytest.raises(KeyError):
        dict_models["NoSuchModel"]
    cfg = ConfigDict(P.cfg_xr1cnn())
    assert cfg["fe"]["arch"] == cfg.fe.arch


def test_schedules_bit_exact():
    from oaprogressionmmf_amd.various import dict_optimizers, dict_schedulers
    tab = json.loads((GOLDEN / "f9_schedules.json").read_text())
    p = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(p, lr=1e-4)
    s = dict_schedulers["CustomWarmupStaticDecayLR"](optimizer=opt, epochs_warmup=5, epochs_static=100, epochs_decay=1)
    lrs = []
    for _ in range(121):
        lrs.append(repr(float(opt.param_groups[0]["lr"])))
        opt.step()
        s.step()
    assert lrs == tab["static_decay"]
    opt = torch.optim.SGD(p, lr=1e-3)
    s = dict_schedulers["CustomWarmupMultiStepLR"](optimizer=opt, epochs_warmup=5, mstep_milestones=[20, 40])
    lrs = []
    for _ in range(121):
        lrs.append(repr(float(opt.param_groups[0]["lr"])))
        opt.step()
        s.step()
    assert lrs == tab["multistep"]
    assert dict_optimizers["Adam"].__module__.startswith("oaprogressionmmf_amd")


def test_no_cpu_fallback_and_no_oracle_in_product():
    from oaprogressionmmf_amd import ops
    from oaprogressionmmf_amd._lib import KoafError
    from oaprogressionmmf_amd.config import ConfigDict
    from oaprogressionmmf_amd.models import dict_models
    from oaprogressionmmf_amd.various import dict_losses
    with pytest.raises(KoafError):
        ops.gelu_fwd(torch.zeros(8))
    m = dict_models["XR1Cnn"](config=ConfigDict(P.cfg_xr1cnn(arch="resnet18", size=64)), path_weights=None)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 64, 64))
    with pytest.raises(KoafError):
        dict_losses["FocalLoss"]()(torch.zeros(2, 2), torch.zeros(2, dtype=torch.int64))
    # the product package never references the oracle
    for f in (ROOT / "oaprogressionmmf_amd").rglob("*.py"):
        txt = f.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt and "koafusion_cpu" not in txt, f


def test_checkpoint_roundtrip(tmp_path):
    from oaprogressionmmf_amd.config import ConfigDict
    from oaprogressionmmf_amd.models import dict_models
    from oaprogressionmmf_amd.various import CheckpointHandler
    cfg = P.cfg_xr1cnn(arch="resnet18", size=64)
    m = dict_models["XR1Cnn"](config=ConfigDict(cfg), path_weights=None)
    h = CheckpointHandler(tmp_path)
    h.save_new_ckpt(model=m, model_name="XR1Cnn", fold_idx=0, epoch_idx=3)
    h.save_new_ckpt(model=m, model_name="XR1Cnn", fold_idx=0, epoch_idx=7)
    files = sorted(p.name for p in tmp_path.glob("*.pth"))
    assert files == ["XR1Cnn__fold_0__epoch_007.pth"]            # num_saved = 1, reference file-name pattern
    cfg2 = dict(cfg, restore_weights=True)
    m2 = dict_models["XR1Cnn"](config=ConfigDict(cfg2), path_weights=h.get_last_ckpt())
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    sd = torch.load(h.get_last_ckpt())
    assert all(v.is_contiguous() for v in sd.values())           # plain tensors: loadable by the reference


def test_checkpoint_newest_save_is_never_the_victim(tmp_path):
    """age is save order, not name order: epoch 1000 sorts before epoch 999 by name, and a run directory may hold another
    model's files (reference: the handler appends to its list, koafusion/various/_checkpoint.py:51-62)"""
    from oaprogressionmmf_amd.various import CheckpointHandler
    m = torch.nn.Linear(2, 2)
    (tmp_path / "ZZ_other__fold_0__epoch_001.pth").write_bytes(b"x")
    h = CheckpointHandler(tmp_path, num_saved=1)
    a = h.save_new_ckpt(model=m, model_name="M", fold_idx=0, epoch_idx=999)
    assert h.get_last_ckpt() == a and not (tmp_path / "ZZ_other__fold_0__epoch_001.pth").exists()
    b = h.save_new_ckpt(model=m, model_name="M", fold_idx=0, epoch_idx=1000)
    assert b.exists() and not a.exists() and h.get_last_ckpt() == b
    c = h.save_new_ckpt(model=m, model_name="A", fold_idx=1, epoch_idx=5)        # sorts first by name, is the newest
    assert c.exists() and not b.exists() and h.get_last_ckpt() == c
    h2 = CheckpointHandler(tmp_path, num_saved=2)
    d = h2.save_new_ckpt(model=m, model_name="M", fold_idx=0, epoch_idx=2)
    assert sorted(p.name for p in tmp_path.glob("*.pth")) == sorted([c.name, d.name]) and h2.get_last_ckpt() == d


def test_checkpoint_resume_orders_foreign_files_by_name_not_mtime(tmp_path):
    """a FRESH handler (resume / eval, koafusion/run/eval_prog_fus.py:161) over a directory whose modification times were
    not preserved: the reference's sorted(glob) order decides (koafusion/various/_checkpoint.py:29,44-46) -- the newest
    epoch by name is the last checkpoint and the older one is trimmed, whatever the mtimes say"""
    import os
    from oaprogressionmmf_amd.various import CheckpointHandler
    old = tmp_path / "M__fold_0__epoch_003.pth"
    new = tmp_path / "M__fold_0__epoch_007.pth"
    new.write_bytes(b"n")
    old.write_bytes(b"o")
    os.utime(new, ns=(1_000_000_000, 1_000_000_000))          # the newer epoch carries the OLDER mtime (cp without -p, checkout)
    os.utime(old, ns=(2_000_000_000, 2_000_000_000))
    h = CheckpointHandler(tmp_path, num_saved=2)
    assert h.get_last_ckpt() == new
    h1 = CheckpointHandler(tmp_path)                            # num_saved = 1 trims from the oldest end = by name
    assert h1.get_last_ckpt() == new and new.exists() and not old.exists()


def test_adamw_takes_capturable():
    from oaprogressionmmf_amd.various import dict_optimizers
    p = torch.nn.Parameter(torch.zeros(3))
    opt = dict_optimizers["AdamW"]([p], lr=1e-3, capturable=True)
    assert opt.capturable and opt._ADAMW
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="arena parameters only"):
        opt.step()                   # a parameter outside an arena cannot take a device-resident step count


def test_fused_adam_amsgrad_state_round_trip():
    """Adam(amsgrad=True): the registry accepts it like torch's, and torch.optim.Adam's state (with max_exp_avg_sq) loads and
    comes back unchanged -- no device involved"""
    from oaprogressionmmf_amd.various import dict_optimizers
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(4, 3, 3, 3)), torch.nn.Parameter(torch.randn(7))]
    ref = torch.optim.Adam(ps, lr=3e-4, amsgrad=True)
    for p in ps:
        p.grad = torch.randn_like(p)
    ref.step()
    sd = ref.state_dict()
    mine = dict_optimizers["Adam"](ps, lr=1.0, amsgrad=True)
    mine.load_state_dict(sd)
    out = mine.state_dict()
    assert out["param_groups"][0]["amsgrad"] is True
    for k, st in sd["state"].items():
        for name in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            assert torch.equal(out["state"][k][name], st[name]), (k, name)
        assert float(out["state"][k]["step"]) == float(st["step"])


def test_fused_adam_state_dict_is_torch_layout():
    """resume bookkeeping without a device: a torch.optim.Adam state_dict loads into the fused Adam and comes back
    unchanged (same keys, steps, moment tensors, hyper-parameters); nothing is computed on the CPU"""
    from oaprogressionmmf_amd.various import dict_optimizers
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(4, 3, 3, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 5))]
    ref = torch.optim.Adam(ps, lr=3e-4, weight_decay=1e-4)
    for p in ps[:2]:                      # the third parameter never receives a gradient -> no state entry
        p.grad = torch.randn_like(p)
    ref.step()
    ref.step()
    sd = ref.state_dict()
    mine = dict_optimizers["Adam"](ps, lr=1.0)
    mine.load_state_dict(sd)
    out = mine.state_dict()
    assert out["param_groups"][0]["lr"] == 3e-4 and out["param_groups"][0]["weight_decay"] == 1e-4
    assert sorted(out["state"]) == sorted(sd["state"]) == [0, 1]
    for k in sd["state"]:
        assert float(out["state"][k]["step"]) == float(sd["state"][k]["step"]) == 2.0
        assert torch.equal(out["state"][k]["exp_avg"], sd["state"][k]["exp_avg"])
        assert torch.equal(out["state"][k]["exp_avg_sq"], sd["state"][k]["exp_avg_sq"])
    back = torch.optim.Adam(ps, lr=1.0)
    back.load_state_dict(out)             # and torch accepts what the fused optimizer exports
    with pytest.raises(RuntimeError):
        mine.step()                       # CPU parameters: refused, no fallback
    with pytest.raises(ValueError):
        dict_optimizers["Adam"](ps[:2], lr=1.0).load_state_dict(sd)


def test_batch_augment_draws_like_the_reference_transforms():
    """host RNG bookkeeping of PTBatchAugment: per sample, rotation (p, theta) then gamma (p, gamma) from Python's
    `random`, exactly the calls the reference's randomize() methods make in list order (oai/_dataset.py:316-321,
    _pt.py:228-232,305-307); absent transforms draw nothing"""
    import math
    import random
    from oaprogressionmmf_amd.preproc import PTBatchAugment
    aug = PTBatchAugment(mean=0.5, std=0.2)
    random.seed(7)
    got = aug.draw(3)
    random.seed(7)
    lo, hi = math.radians(-15.), math.radians(15.)
    want = []
    for _ in range(3):
        p1 = random.random(); th = random.uniform(lo, hi); p2 = random.random(); g = random.uniform(0.5, 2.0)
        want.append((p1, th, p2, g))
    assert got == want
    t2 = PTBatchAugment(mean=0.259, std=0.345, gamma_prob=0.0)
    random.seed(7)
    a = t2.draw(2)
    random.seed(7)
    b = [(random.random(), random.uniform(lo, hi), 1.0, 1.0) for _ in range(2)]
    assert a == b
    random.seed(7)
    s0 = random.getstate()
    assert PTBatchAugment(mean=0., std=1., rotate_prob=0.0, gamma_prob=0.0).draw(4) == [(1.0, 0.0, 1.0, 1.0)] * 4
    assert random.getstate() == s0
    with pytest.raises(NotImplementedError):
        PTBatchAugment(mean=0., std=1., clip_to_unit=True)


def test_bench_workloads_are_constructible():
    """bench.py's workload table: every name yields a config the registry accepts (constructed on the CPU, nothing
    computed), has an algorithmic-FLOP entry, and -- where a CPU baseline is taken -- one the oracle can state"""
    import importlib.util
    from oracle import koafusion_cpu as O
    from oaprogressionmmf_amd.config import ConfigDict
    from oaprogressionmmf_amd.models import dict_models
    spec = importlib.util.spec_from_file_location("koaf_bench", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    names = ("native3", "eval3", "native", "syn", "syn3", "xr1cnn", "xr1c1", "mr1", "mr1c1")
    for name in names:
        cfg, b, policy = bench.workload_cfg(name)
        assert b >= 1 and isinstance(policy, str) and bench.algorithmic_train_gflop_per_sample(name) > 0
        shapes = cfg.pop("_tensor_shapes", None)
        if name not in ("syn", "syn3", "mr1c1", "native3", "eval3", "native"):        # (the big ones: ~0.4-0.6 G parameters each)
            m = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None)
            spec_o, _ = O.model_spec(cfg)
            assert sorted(k for k, _ in m.state_dict().items()) == sorted(k for k, _, _ in spec_o)
        if shapes is not None:
            assert len(shapes) == len(cfg["input_size"])
        ins = P.model_inputs(dict(cfg, input_size=[[8, 8] if len(s) == 2 else ([8, 8, 2] if len(s) == 3 else s)
                                                  for s in cfg["input_size"]]), 1)
        assert len(ins) == len(cfg["input_size"])
    with pytest.raises(SystemExit):
        bench.workload_cfg("nope")


def test_pretrained_encoders_load_from_the_local_hub_cache(tmp_path, monkeypatch):
    """`fe.*.pretrained: true` -- what every shipped reference recipe sets (runner.sh:94,116,..., conf/model/*.yaml) -- resolves
    the file the reference's load_state_dict_from_url would have cached (koafusion/models/_torchvision.py:249-261) instead of a
    download: $KOAF_PRETRAINED_DIR or <TORCH_HOME>/hub/checkpoints; strict load (the keys incl. `fc` are the reference's);
    absent file = RuntimeError, never a silent random init"""
    from oaprogressionmmf_amd.config import load_model_config
    from oaprogressionmmf_amd.models import dict_models
    from oaprogressionmmf_amd.models._core_fes import dict_fes, model_urls, pretrained_candidates
    monkeypatch.delenv("KOAF_PRETRAINED_DIR", raising=False)
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "th"))
    with pytest.raises(RuntimeError, match="resnet18-f37072fd.pth"):
        dict_fes["resnet18"](pretrained=True)
    # a stand-in "ImageNet" checkpoint with the reference's key set (values by key name)
    def ckpt(arch):
        sd = dict_fes[arch](pretrained=False).state_dict()
        return {k: torch.from_numpy(P.fill_value(k, tuple(v.shape), v.dtype == torch.int64)).to(v.dtype).reshape(v.shape) for k, v in sd.items()}
    hub = tmp_path / "th" / "hub" / "checkpoints"
    hub.mkdir(parents=True)
    sd18 = ckpt("resnet18")
    torch.save(sd18, hub / "resnet18-f37072fd.pth")
    assert pretrained_candidates("resnet18")[-1] == hub / "resnet18-f37072fd.pth"
    m = dict_fes["resnet18"](pretrained=True)
    assert all(torch.equal(v, sd18[k]) for k, v in m.state_dict().items()) and "fc.weight" in sd18
    # $KOAF_PRETRAINED_DIR wins over the hub cache; a whole registry model built from a config with pretrained: true
    other = tmp_path / "ckpts"
    other.mkdir()
    sd18b = {k: (v + 1 if v.dtype.is_floating_point else v) for k, v in sd18.items()}
    torch.save(sd18b, other / "resnet18-f37072fd.pth")
    monkeypatch.setenv("KOAF_PRETRAINED_DIR", str(other))
    cfg = load_model_config("xr1_cnn", **{"fe.arch": "resnet18", "fe.pretrained": True})
    model = dict_models["XR1Cnn"](config=cfg, path_weights=None)
    got = model.state_dict()
    assert torch.equal(got["_fe.0.weight"], sd18b["conv1.weight"]) and torch.equal(got["_fe.4.0.bn1.running_var"], sd18b["layer1.0.bn1.running_var"])
    assert set(model_urls) == {"resnet18", "resnet34", "resnet50", "resnext50_32x4d"}
    with pytest.raises(RuntimeError, match="resnet50-0676ba61.pth"):        # the MRI trunks' file is still absent
        dict_fes["resnet50"](pretrained=True)


def test_model_config_groups_load():
    """every conf/model/*.yaml names a registry model and carries the keys its constructor reads"""
    from oaprogressionmmf_amd.config import CONF_DIR, load_model_config
    from oaprogressionmmf_amd.models import dict_models
    names = sorted(p.stem for p in (CONF_DIR / "model").glob("*.yaml"))
    assert {"xr1_cnn", "mr1_cnn_trf", "mr2_cnn_trf", "xr1mr1_cnn_trf", "xr1mr2_cnn_trf", "xr1mr2c1_cnn_trf",
            "xr1c1_cnn", "mr1c1_cnn_trf", "xr1mr3c1_cnn_trf"} <= set(names)
    for n in names:
        cfg = load_model_config(n)
        assert cfg["name"] in dict_models and len(cfg["input_size"]) == len(cfg["downscale"])
        ns = cfg["agg"].get("num_slices") if hasattr(cfg["agg"], "get") else None
        if isinstance(ns, (list, tuple)):
            assert len(ns) == len(cfg["input_size"])
    m = dict_models["XR1C1Cnn"](config=load_model_config("xr1c1_cnn"), path_weights=None)
    assert m.vs["agg_in_len"] == 4096


def test_clock_per_kernel_reduction(tmp_path):
    """scripts/clock_per_kernel.py: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, matrix-pipe share = busy cycles / 1024 SIMDs / cycles,
    per kernel template, first two launches of a group skipped"""
    import json
    import subprocess
    import sys
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    name = '"void (anonymous namespace)::koaf_gemm_kernel<256, 128, 9, 6, 0, 0, true, true, 512, 0, false, 0>(KoafGemm)"'
    for d in range(1, 8):
        dur_ns = 2_000_000 if d > 2 else 9_000_000                    # (the two skipped launches would spoil the average)
        cyc = 1.9e9 * dur_ns * 1e-9                                  # 1.9 GHz
        for xcd in range(8):
            rows.append(f"{d},{name},GRBM_GUI_ACTIVE,{cyc},{1000},{1000 + dur_ns}")
        rows.append(f"{d},{name},SQ_VALU_MFMA_BUSY_CYCLES,{0.5 * cyc * 1024},{1000},{1000 + dur_ns}")
    src = tmp_path / "c.csv"
    src.write_text("\n".join(rows) + "\n")
    out = tmp_path / "o.json"
    root = Path(__file__).resolve().parent.parent
    subprocess.run([sys.executable, str(root / "scripts" / "clock_per_kernel.py"), str(src), str(out), "1"], check=True, capture_output=True)
    k = json.loads(out.read_text())["kernels"]
    assert len(k) == 1 and k[0]["launches"] == 5 and k[0]["kernel"].startswith("koaf_gemm_kernel<256, 128, 9,")
    assert abs(k[0]["clock_ghz"] - 1.9) < 1e-3 and abs(k[0]["mfma_busy_frac_at_clock"] - 0.5) < 1e-3 and abs(k[0]["avg_us"] - 2000.0) < 1e-6
