"""CPU: the evaluation-regime host logic.
koafusion/run/eval_prog_fus.py cannot be imported here (thop/captum/cv2/hydra are absent), so the oracle's
`ensemble_foldw` is pinned against the published algorithm run with the same third-party calls the reference
makes (pandas.merge(validate="1:1") + scipy.special.softmax, eval_prog_fus.py:317-343) and against a
hand-derived known answer; the product's ensemble_eval_foldw must then equal the oracle exactly."""
import functools
import math
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import koafusion_cpu as O  # noqa: E402
from oaprogressionmmf_amd.run import ensemble_eval_foldw  # noqa: E402


def _folds(n=37, nfold=5, ncls=2, seed=3, drop=True):
    rng = np.random.default_rng(seed)
    ids = [f"knee_{i:04d}" for i in range(n)]
    tgt = rng.integers(0, ncls, n).tolist()
    raw = {}
    for k in range(nfold):
        order = rng.permutation(n)
        if drop and k in (1, 3):
            order = order[: n - 3 - k]                       # some exams missing in some folds
        logits = rng.normal(size=(len(order), ncls)) * 2
        p = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
        raw[k] = dict(exam_knee_id=[ids[i] for i in order], target=[tgt[i] for i in order],
                      predict=np.argmax(p, -1).tolist(), predict_proba=p.astype(np.float32).tolist())
    return raw


def _published_algorithm(raw_foldw):
    """the reference's statement sequence with its own library calls"""
    import pandas as pd
    from scipy.special import softmax
    dfs = []
    for k, d in raw_foldw.items():
        t = pd.DataFrame.from_dict(d)
        dfs.append(t.rename(columns={"predict": f"predict__{k}", "predict_proba": f"predict_proba__{k}"}))
    dfs[1:] = [e.drop(columns="target") for e in dfs[1:]]
    df = functools.reduce(lambda l, r: pd.merge(l, r, on=["exam_knee_id"], validate="1:1"), dfs)
    cols = [c for c in df.columns if c.startswith("predict_proba__")]
    t = softmax(np.mean(np.asarray(df[cols].values.tolist()), axis=1), axis=-1)
    df["predict_proba"] = t.tolist()
    df["predict"] = np.argmax(t, axis=-1).tolist()
    return df.to_dict(orient="list")


@pytest.mark.parametrize("ncls,drop", [(2, True), (2, False), (3, True)])
def test_oracle_ensemble_matches_published_algorithm(ncls, drop):
    raw = _folds(ncls=ncls, drop=drop)
    want = _published_algorithm(raw)
    got = O.ensemble_foldw(raw)
    assert list(got.keys()) == list(want.keys())
    for k in want:
        if k.startswith("predict_proba"):
            np.testing.assert_allclose(np.asarray(got[k]), np.asarray(want[k]), rtol=0, atol=1e-15)
        else:
            assert got[k] == want[k], k


def test_known_answer():
    raw = {0: dict(exam_knee_id=["a", "b"], target=[1, 0], predict=[0, 1], predict_proba=[[0.8, 0.2], [0.1, 0.9]]),
           4: dict(exam_knee_id=["b", "a"], target=[0, 1], predict=[1, 0], predict_proba=[[0.3, 0.7], [0.6, 0.4]])}
    for fn in (O.ensemble_foldw, ensemble_eval_foldw):
        ens = fn(raw)
        assert ens["exam_knee_id"] == ["a", "b"] and ens["target"] == [1, 0]
        assert ens["predict__4"] == [0, 1] and ens["predict_proba__4"] == [[0.6, 0.4], [0.3, 0.7]]
        # mean a = (0.7, 0.3), b = (0.2, 0.8); softmax over the probabilities
        pa = math.exp(0.7) / (math.exp(0.7) + math.exp(0.3))
        pb = math.exp(0.2) / (math.exp(0.2) + math.exp(0.8))
        np.testing.assert_allclose(ens["predict_proba"], [[pa, 1 - pa], [pb, 1 - pb]], rtol=1e-14)
        assert ens["predict"] == [0, 1]


@pytest.mark.parametrize("ncls,drop", [(2, True), (3, False)])
def test_product_ensemble_equals_oracle(ncls, drop):
    raw = _folds(ncls=ncls, drop=drop, seed=11)
    want, got = O.ensemble_foldw(raw), ensemble_eval_foldw(raw)
    assert list(got.keys()) == list(want.keys())
    for k in want:
        if k == "predict_proba":
            np.testing.assert_allclose(np.asarray(got[k]), np.asarray(want[k]), rtol=0, atol=1e-15)
        else:
            assert got[k] == want[k], k


def test_ensemble_edge_cases():
    one = {2: _folds(nfold=1)[0]}
    ens = ensemble_eval_foldw(one)                                   # single fold: still softmax(proba)
    assert ens["exam_knee_id"] == one[2]["exam_knee_id"] and len(ens["predict"]) == len(ens["exam_knee_id"])
    dup = _folds(nfold=2, drop=False)
    dup[1]["exam_knee_id"][0] = dup[1]["exam_knee_id"][1]
    for fn in (O.ensemble_foldw, ensemble_eval_foldw, _published_algorithm):
        with pytest.raises(Exception):
            fn(dup)
    disjoint = {0: dict(exam_knee_id=["a"], target=[0], predict=[0], predict_proba=[[0.5, 0.5]]),
                1: dict(exam_knee_id=["b"], target=[0], predict=[0], predict_proba=[[0.5, 0.5]])}
    ens = ensemble_eval_foldw(disjoint)
    assert ens["exam_knee_id"] == [] and ens["predict"] == [] and ens["predict_proba"] == []
    assert O.ensemble_foldw(disjoint)["predict"] == []
    with pytest.raises(ValueError):
        ensemble_eval_foldw({})


def test_eval_epoch_refuses_cpu_and_unknown_profile():
    import torch
    from oaprogressionmmf_amd.run import eval_epoch
    with pytest.raises(ValueError):
        eval_epoch(None, [], ("xr_pa",), profile="compute")
    assert eval_epoch(None, [], ("xr_pa",)) == {}
    with pytest.raises(AssertionError):
        eval_epoch(None, [{"image__xr_pa": torch.zeros(1)}], ("bogus",), device="cpu")
