"""Framework-free (NumPy) procedural weights, inputs and result summaries shared by
  * tests/golden/make_golden.py (runs the imported reference in the build container),
  * the oracle tests (CPU) and the HIP parity tests (GPU box, where /root/reference does not exist).
Every state_dict entry is filled BY KEY NAME, so all three sides hold bit-identical parameters without
any weight file travelling.
"""
import zlib

import numpy as np


def _rng(key, salt=0):
    return np.random.default_rng(zlib.crc32(key.encode()) + 1000003 * salt)


_FILL_CACHE, _FILL_BYTES, _FILL_CAP = {}, [0], 6 << 30


def fill_value(key, shape, is_int=False):
    """Deterministic value for state_dict entry `key` (numpy array, float32 or int64).  The generated arrays are cached
    (the product model and the float32 / float64 oracles of one test ask for the same ones); callers get a copy."""
    ck = (key, tuple(shape), bool(is_int))
    hit = _FILL_CACHE.get(ck)
    if hit is None:
        hit = _fill_value(key, shape, is_int)
        if _FILL_BYTES[0] + hit.nbytes > _FILL_CAP:
            _FILL_CACHE.clear()
            _FILL_BYTES[0] = 0
        _FILL_CACHE[ck] = hit
        _FILL_BYTES[0] += hit.nbytes
    return hit.copy()


def _fill_value(key, shape, is_int=False):
    shape = tuple(shape)
    if is_int or key.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    r = _rng(key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "running_var":
        return r.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if leaf == "running_mean":
        return r.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    if leaf in ("cls_token", "pos_embedding"):
        return r.uniform(-1.0, 1.0, size=shape).astype(np.float32)
    if len(shape) == 1:
        if leaf == "weight":      # BatchNorm / LayerNorm scale
            return r.uniform(0.9, 1.1, size=shape).astype(np.float32)
        return r.uniform(-0.05, 0.05, size=shape).astype(np.float32)   # biases
    fan_in = int(np.prod(shape[1:]))
    a = np.sqrt(3.0 / fan_in)
    return r.uniform(-a, a, size=shape).astype(np.float32)


def imagenet_like_fill(key, shape, is_int=False):
    """Deterministic "ImageNet-checkpoint-like" value for state_dict entry `key`: the parameter DISTRIBUTION of a trained
    torchvision ResNet rather than a fresh init -- the real files cannot travel (no network), and the fixtures' gentle fill
    (BatchNorm scale 1 +- 0.1, variance 0.5 .. 1.5) says nothing about the fp16-piece scheme's fixed activation scale on such
    weights.  BatchNorm scale log-uniform over 1e-3 .. 3 with a few exact zeros and a few negative entries (trained nets prune
    channels this way), bias N(0, 0.3), running variance log-uniform over 1e-4 .. 10, running mean N(0, 0.5) * sqrt(var);
    convolution weights heavy-tailed (Student-t, 4 degrees of freedom) with a per-layer gain spread over two decades."""
    shape = tuple(shape)
    if is_int or key.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    r = _rng("imagenet:" + key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "running_var":
        return np.exp(r.uniform(np.log(1e-4), np.log(10.0), size=shape)).astype(np.float32)
    if leaf == "running_mean":
        var = np.exp(_rng("imagenet:" + key[:-len("running_mean")] + "running_var").uniform(np.log(1e-4), np.log(10.0), size=shape))
        return (r.standard_normal(size=shape) * 0.5 * np.sqrt(var)).astype(np.float32)
    if len(shape) == 1:
        if leaf == "weight":
            g = np.exp(r.uniform(np.log(1e-3), np.log(3.0), size=shape))
            u = r.uniform(size=shape)
            g = np.where(u < 0.04, 0.0, np.where(u < 0.09, -g, g))
            return g.astype(np.float32)
        return (0.3 * r.standard_normal(size=shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    gain = 10.0 ** r.uniform(-1.5, 0.5)
    w = r.standard_t(4, size=shape) / np.sqrt(2.0)          # (variance of t_4 is 2)
    return (gain * np.sqrt(2.0 / fan_in) * w).astype(np.float32)


def fill_state_dict(sd, fill=None):
    """In-place fill of a torch state_dict (values must be tensors) -- returns sd.  fill: fill_value (default) or another
    `(key, shape, is_int) -> ndarray` law, e.g. imagenet_like_fill"""
    import torch
    for k, v in sd.items():
        arr = (fill or fill_value)(k, v.shape, is_int=not v.dtype.is_floating_point)
        v.copy_(torch.from_numpy(arr).to(v.dtype).reshape(v.shape))
    return sd


def make_input(name, shape, seed=1234):
    r = _rng(name, salt=seed)
    return r.standard_normal(size=tuple(shape)).astype(np.float32)


def make_clin(name, B, seed=1234):
    """[z, onehot2, z, onehot2, onehot2, z] per koafusion/datasets/oai/_dataset.py:254-266 -> (B,1,9)"""
    r = _rng(name, salt=seed)
    out = np.zeros((B, 1, 9), dtype=np.float32)
    for b in range(B):
        z = r.standard_normal(3)
        oh = r.integers(0, 2, size=3)
        v = [z[0], 1 - oh[0], oh[0], z[1], 1 - oh[1], oh[1], 1 - oh[2], oh[2], z[2]]
        out[b, 0] = np.asarray(v, dtype=np.float32)
    return out


def make_target(name, B, seed=1234):
    r = _rng(name, salt=seed)
    return r.integers(0, 2, size=(B, 1)).astype(np.int64)


def sample_index(key, numel, k=8):
    r = _rng("idx:" + key)
    return np.sort(r.integers(0, numel, size=min(k, numel)))


def summarize_tensors(named, k=8):
    """{name: tensor/array} -> {name+':norm': float64, name+':samples': float32[8]} (flattened C-order of the
    LOGICAL shape, so memory layout differences do not matter)."""
    out = {}
    for key, v in named.items():
        a = np.asarray(v, dtype=np.float64).reshape(-1)
        out[key + ":norm"] = np.float64(np.sqrt((a * a).sum()))
        out[key + ":samples"] = a[sample_index(key, a.size, k)].astype(np.float32)
    return out


# ---------------------------------------------------------------------------------------------
# fixture model configurations (plain dicts; wrapped by each side's own attr+item config class)
# ---------------------------------------------------------------------------------------------
def cfg_xr1cnn(arch="resnext50_32x4d", size=350, dropout=0.0):
    return dict(name="XR1Cnn", input_size=[[size, size]], downscale=False, input_channels=1, output_channels=2,
                fe=dict(arch=arch, pretrained=False, with_gap=True, dropout=0.0),
                agg=dict(hidden_size=512, dropout=dropout), output_type="dict", pretrained=False,
                path_pretrained=None, restore_weights=False, debug=False)


def cfg_mr1(arch="resnet50", shape=(160, 160, 64), dims_view="rc", with_gap=True, depth=4, heads=8, dropout=0.0):
    return dict(name="MR1CnnTrf", input_size=[list(shape)], downscale=False, input_channels=1, output_channels=2,
                fe=dict(arch=arch, pretrained=False, with_gap=with_gap, dropout=0.0, dims_view=dims_view),
                agg=dict(num_slices=shape[2], depth=depth, heads=heads, emb_dropout=dropout, mlp_dim=2048,
                         mlp_dropout=dropout),
                output_type="dict", pretrained=False, path_pretrained=None, restore_weights=False, debug=False)


def cfg_mr2(shape0=(160, 160, 8), shape1=(160, 160, 6), depth=1):
    return dict(name="MR2CnnTrf", input_size=[list(shape0), list(shape1)], downscale=False, input_channels=1,
                output_channels=2, fe=dict(arch="resnet50", pretrained=False, with_gap=True, dropout=0.0),
                agg=dict(num_slices=[shape0[2], shape1[2]], depth=depth, heads=8, emb_dropout=0.0, mlp_dim=2048,
                         mlp_dropout=0.0),
                output_type="dict", pretrained=False, path_pretrained=None, restore_weights=False, debug=False)


def cfg_xr1mr1(xr=(160, 160), mr=(160, 160, 6), depth=1, xr_arch="resnext50_32x4d", with_gap=True):
    return dict(name="XR1MR1CnnTrf", input_size=[list(xr), list(mr)], downscale=False, input_channels=1,
                output_channels=2,
                fe=dict(xr=dict(arch=xr_arch, pretrained=False, with_gap=with_gap, dropout=0.0),
                        mr=dict(arch="resnet50", pretrained=False, with_gap=with_gap, dropout=0.0)),
                agg=dict(num_slices=[1, mr[2]], depth=depth, heads=8, emb_dropout=0.0, mlp_dim=2048, mlp_dropout=0.0),
                output_type="dict", pretrained=False, path_pretrained=None, restore_weights=False, debug=False)


def cfg_xr1mr2(xr=(160, 160), mr1=(160, 160, 6), mr2=(160, 160, 5), depth=1, xr_arch="resnext50_32x4d"):
    return dict(name="XR1MR2CnnTrf", input_size=[list(xr), list(mr1), list(mr2)], downscale=False,
                input_channels=1, output_channels=2,
                fe=dict(xr=dict(arch=xr_arch, pretrained=False, with_gap=True, dropout=0.0),
                        mr=dict(arch="resnet50", pretrained=False, with_gap=True, dropout=0.0)),
                agg=dict(num_slices=[1, mr1[2], mr2[2]], depth=depth, heads=8, emb_dropout=0.0, mlp_dim=2048,
                         mlp_dropout=0.0),
                output_type="dict", pretrained=False, path_pretrained=None, restore_weights=False, debug=False)


def cfg_full(xr=(350, 350), mr1=(160, 160, 64), mr2=(160, 160, 25), depth=4, dropout=0.0,
             xr_arch="resnext50_32x4d"):
    """XR1MR2C1CnnTrf as runner.sh:347-361 (inputs already at model size: downscale false)"""
    return dict(name="XR1MR2C1CnnTrf", input_size=[list(xr), list(mr1), list(mr2), [16]], downscale=False,
                input_channels=1, output_channels=2,
                fe=dict(xr=dict(arch=xr_arch, pretrained=False, with_gap=True, dropout=dropout),
                        mr=dict(arch="resnet50", pretrained=False, with_gap=True, dropout=dropout),
                        clin=dict(dim_in=9, dim_out=2048, dropout=dropout)),
                agg=dict(num_slices=[1, mr1[2], mr2[2], 1], depth=depth, heads=8, emb_dropout=dropout,
                         mlp_dim=2048, mlp_dropout=dropout),
                output_type="dict", pretrained=False, path_pretrained=None, restore_weights=False, debug=False)


def cfg_xr1c1(arch="resnext50_32x4d", size=350, dropout=0.0):
    """extension XR1C1Cnn (BASELINE config "XR-PA + clinical early-fusion MLP head")"""
    return dict(name="XR1C1Cnn", input_size=[[size, size], [16]], downscale=False, input_channels=1,
                output_channels=2,
                fe=dict(xr=dict(arch=arch, pretrained=False, with_gap=True, dropout=dropout),
                        clin=dict(dim_in=9, dim_out=2048, dropout=dropout)),
                agg=dict(hidden_size=512, dropout=dropout), output_type="dict", pretrained=False,
                path_pretrained=None, restore_weights=False, debug=False)


def cfg_mr1c1(mr=(160, 160, 64), depth=4, dropout=0.0):
    """extension MR1C1CnnTrf (BASELINE config "SAG-3D-DESS encoder + clinical")"""
    return dict(name="MR1C1CnnTrf", input_size=[list(mr), [16]], downscale=False, input_channels=1,
                output_channels=2,
                fe=dict(mr=dict(arch="resnet50", pretrained=False, with_gap=True, dropout=dropout),
                        clin=dict(dim_in=9, dim_out=2048, dropout=dropout)),
                agg=dict(num_slices=[mr[2], 1], depth=depth, heads=8, emb_dropout=dropout, mlp_dim=2048,
                         mlp_dropout=dropout),
                output_type="dict", pretrained=False, path_pretrained=None, restore_weights=False, debug=False)


def cfg_xr1mr3c1(xr=(350, 350), mr1=(160, 160, 64), mr2=(160, 160, 32), mr3=(160, 160, 25), depth=4, dropout=0.0,
                 xr_arch="resnext50_32x4d"):
    """extension XR1MR3C1CnnTrf (BASELINE config "Full XR + SAG-DESS/COR-IW-TSE/SAG-T2 + clinical")"""
    return dict(name="XR1MR3C1CnnTrf", input_size=[list(xr), list(mr1), list(mr2), list(mr3), [16]],
                downscale=False, input_channels=1, output_channels=2,
                fe=dict(xr=dict(arch=xr_arch, pretrained=False, with_gap=True, dropout=dropout),
                        mr=dict(arch="resnet50", pretrained=False, with_gap=True, dropout=dropout),
                        clin=dict(dim_in=9, dim_out=2048, dropout=dropout)),
                agg=dict(num_slices=[1, mr1[2], mr2[2], mr3[2], 1], depth=depth, heads=8, emb_dropout=dropout,
                         mlp_dim=2048, mlp_dropout=dropout),
                output_type="dict", pretrained=False, path_pretrained=None, restore_weights=False, debug=False)


def model_inputs(cfg, B, seed=1234):
    """numpy inputs in the model's positional order"""
    name = cfg["name"]
    sz = cfg["input_size"]
    if name == "XR1Cnn":
        return [make_input("xr", (B, 1, *sz[0]), seed)]
    if name == "MR1CnnTrf":
        return [make_input("mr0", (B, 1, *sz[0]), seed)]
    if name == "MR2CnnTrf":
        return [make_input("mr0", (B, 1, *sz[0]), seed), make_input("mr1", (B, 1, *sz[1]), seed)]
    if name == "XR1MR1CnnTrf":
        return [make_input("xr", (B, 1, *sz[0]), seed), make_input("mr0", (B, 1, *sz[1]), seed)]
    if name == "XR1MR2CnnTrf":
        return [make_input("xr", (B, 1, *sz[0]), seed), make_input("mr0", (B, 1, *sz[1]), seed),
                make_input("mr1", (B, 1, *sz[2]), seed)]
    if name == "XR1C1Cnn":
        return [make_input("xr", (B, 1, *sz[0]), seed), make_clin("clin", B, seed)]
    if name == "MR1C1CnnTrf":
        return [make_input("mr0", (B, 1, *sz[0]), seed), make_clin("clin", B, seed)]
    if name == "XR1MR3C1CnnTrf":
        return [make_input("xr", (B, 1, *sz[0]), seed), make_input("mr0", (B, 1, *sz[1]), seed),
                make_input("mr1", (B, 1, *sz[2]), seed), make_input("mr2", (B, 1, *sz[3]), seed),
                make_clin("clin", B, seed)]
    if name == "XR1MR2C1CnnTrf":
        return [make_input("xr", (B, 1, *sz[0]), seed), make_input("mr0", (B, 1, *sz[1]), seed),
                make_input("mr1", (B, 1, *sz[2]), seed), make_clin("clin", B, seed)]
    raise KeyError(name)
