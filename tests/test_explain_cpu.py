"""CPU: the explanation regime (koafusion/run/eval_prog_fus.py:410-512).
 * the oracle's modal_ablation / ablation_percent on the ORACLE model reproduce fixture F13, which holds the imported
   reference model's own logits with each modality zeroed (tests/golden/make_golden.py case_f13_modal_abl);
 * the oracle's ensemble_explain_foldw is pinned against the published statement sequence run with pandas
   (the reference's own third-party calls) and a hand-derived known answer;
 * the product's host logic (ablation_percent, ensemble_explain_foldw) must equal the oracle exactly."""
import functools
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import procedural as P  # noqa: E402
from oracle import koafusion_cpu as O  # noqa: E402
from oaprogressionmmf_amd.run import ablation_percent, ensemble_explain_foldw, explain_epoch  # noqa: E402

GOLD = Path(__file__).resolve().parent / "golden"


def test_oracle_modal_ablation_matches_reference_fixture():
    g = np.load(GOLD / "f13_modal_abl.npz")
    cfg, B, seed = json.loads(str(g["cfg_json"])), int(g["B"]), int(g["seed"])
    torch.set_num_threads(8)
    om = O.OracleModel(cfg, fill=P.fill_value)
    xs = [torch.from_numpy(a) for a in P.model_inputs(cfg, B, seed)]
    y = torch.from_numpy(P.make_target("target", B, seed))
    assert np.array_equal(y.numpy(), g["target"])

    def fwd(*inp):
        out = om(*inp, train=False)
        return out["main"] if isinstance(out, dict) else out
    attrs = O.modal_ablation(fwd, xs, y.squeeze())
    scale = np.abs(g["logits"]).max()
    assert np.abs(attrs.numpy() - g["attrs"]).max() < 5e-6 * max(1.0, scale)
    assert np.abs(O.ablation_percent(attrs) - g["percent"]).max() < 0.02
    # the rule itself on the reference's logits: exact
    sel = np.take_along_axis(g["logits"], np.broadcast_to(g["target"][None], (g["logits"].shape[0], B, 1)), axis=2)[..., 0]
    np.testing.assert_array_equal((sel[0][None] - sel[1:]).T.astype(np.float32), g["attrs"])
    np.testing.assert_array_equal(O.ablation_percent(g["attrs"]), g["percent"])
    np.testing.assert_array_equal(ablation_percent(g["attrs"]), g["percent"])


def _folds(n=23, nfold=4, nmod=4, seed=5, drop=True):
    rng = np.random.default_rng(seed)
    ids = [f"knee_{i:04d}" for i in range(n)]
    tgt = rng.integers(0, 2, (n, 1)).tolist()
    names = ["xr_pa", "sag_3d_dess", "sag_t2_map", "clin"][:nmod]
    raw = {}
    for k in range(nfold):
        order = rng.permutation(n)
        if drop and k in (1, 2):
            order = order[: n - 2 - k]
        attrs = rng.normal(size=(len(order), nmod)).astype(np.float32) * 0.1
        raw[k] = dict(exam_knee_id=[ids[i] for i in order], target=[tgt[i] for i in order],
                      modal_names=[names] * len(order), modal_abl_attrs=attrs.tolist(),
                      modal_abl_percent=O.ablation_percent(attrs).tolist())
    return raw


def _published_algorithm(raw_foldw):
    """the reference's statement sequence (eval_prog_fus.py:481-512) with its own pandas calls"""
    import pandas as pd
    dfs = []
    for k, d in raw_foldw.items():
        t = pd.DataFrame.from_dict(d)
        dfs.append(t.rename(columns={"modal_abl_attrs": f"modal_abl_attrs__{k}",
                                     "modal_abl_percent": f"modal_abl_percent__{k}"}))
    for field in ("target", "modal_names"):
        dfs[1:] = [e.drop(columns=field) for e in dfs[1:]]
    df = functools.reduce(lambda l, r: pd.merge(l, r, on=["exam_knee_id"], validate="1:1"), dfs)
    cols = [c for c in df.columns if c.startswith("modal_abl_percent__")]
    t = np.mean(np.asarray(df[cols].values.tolist()), axis=1)
    df["modal_abl_percent"] = (t / np.sum(t, axis=1, keepdims=True)).tolist()
    return df.to_dict(orient="list")


@pytest.mark.parametrize("nfold,drop", [(4, True), (2, False), (1, False)])
def test_explain_ensemble_oracle_and_product_match_published_algorithm(nfold, drop):
    raw = _folds(nfold=nfold, drop=drop)
    want = _published_algorithm(raw)
    for fn in (O.ensemble_explain_foldw, ensemble_explain_foldw):
        got = fn(raw)
        assert list(got.keys()) == list(want.keys())
        for k in want:
            if k == "modal_abl_percent":
                np.testing.assert_allclose(np.asarray(got[k]), np.asarray(want[k]), rtol=0, atol=1e-15)
            else:
                assert got[k] == want[k], k


def test_explain_known_answer_and_edges():
    raw = {0: dict(exam_knee_id=["a", "b"], target=[[1], [0]], modal_names=[["x", "c"]] * 2,
                   modal_abl_attrs=[[0.3, -0.1], [0.0, 0.2]], modal_abl_percent=[[75.0, 25.0], [0.0, 100.0]]),
           2: dict(exam_knee_id=["b", "a"], target=[[0], [1]], modal_names=[["x", "c"]] * 2,
                   modal_abl_attrs=[[0.1, 0.1], [-0.2, 0.2]], modal_abl_percent=[[50.0, 50.0], [50.0, 50.0]])}
    for fn in (O.ensemble_explain_foldw, ensemble_explain_foldw):
        ens = fn(raw)
        assert ens["exam_knee_id"] == ["a", "b"] and ens["target"] == [[1], [0]]
        assert ens["modal_abl_attrs__2"] == [[-0.2, 0.2], [0.1, 0.1]]
        np.testing.assert_allclose(ens["modal_abl_percent"], [[0.625, 0.375], [0.25, 0.75]], rtol=1e-15)
        disjoint = {0: dict(raw[0]), 1: dict(raw[2], exam_knee_id=["c", "d"])}
        assert fn(disjoint)["modal_abl_percent"] == [] and fn(disjoint)["exam_knee_id"] == []
        dup = {0: dict(raw[0]), 1: dict(raw[2], exam_knee_id=["a", "a"])}
        with pytest.raises(ValueError):
            fn(dup)
    np.testing.assert_array_equal(ablation_percent([[0.3, -0.1], [0.0, 0.2]]),
                                  np.array([[75.0, 25.0], [0.0, 100.0]], dtype=np.float32))
    with pytest.raises(ValueError):
        ensemble_explain_foldw({})
    with pytest.raises(ValueError):
        explain_epoch(None, [], ("xr_pa",), explain_fn="grad_cam")
    assert explain_epoch(None, [], ("xr_pa",)) == {}
