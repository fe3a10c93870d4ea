"""Shared comparison helpers for the oracle-vs-golden (CPU) and HIP-vs-golden (GPU) tests."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def load(name):
    return np.load(GOLDEN / name, allow_pickle=False)


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def check_summary(got, gold, prefix, tol, what=""):
    """Every `prefix...:norm` / `:samples` entry of the golden file against `got` (same key scheme).
    norm: relative; samples: |d| <= tol * (|gold| + rms of the tensor is unknown -> max|samples|)."""
    bad = []
    n = 0
    for k in gold.files if hasattr(gold, "files") else gold:
        if not k.startswith(prefix):
            continue
        if k.endswith(":norm"):
            n += 1
            g, r = float(got[k]), float(gold[k])
            if abs(g - r) > tol * max(abs(r), 1e-12):
                bad.append((k, g, r))
        elif k.endswith(":samples"):
            g, r = np.asarray(got[k], dtype=np.float64), np.asarray(gold[k], dtype=np.float64)
            scale = max(np.abs(r).max(), 1e-12)
            if np.abs(g - r).max() > 4 * tol * scale:
                bad.append((k, g.tolist(), r.tolist()))
    assert n > 0, f"no golden entries under {prefix}"
    assert not bad, f"{what}: {len(bad)}/{n} summaries off (tol {tol}); first: {bad[:3]}"
    return n


def cfg_of(gold):
    return json.loads(str(gold["cfg_json"]))
