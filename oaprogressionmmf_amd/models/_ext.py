"""Registry EXTENSIONS for the BASELINE.json configurations the reference has no class for:

  XR1C1Cnn        C2  "XR-PA + clinical early-fusion MLP head"  = XR1Cnn (koafusion/models/_xr1_cnn.py:9-81) with
                      the FeatC1 embedding of the clinical vector (_xrNmrMcP.py:11-29) concatenated to the
                      pooled trunk features in front of the `_agg` MLP
  MR1C1CnnTrf     C3  "SAG-3D-DESS encoder + clinical"          = the hierarchical pattern of
                      XR1MR2C1CnnTrf (_xrNmrMcP.py:32-264) with no radiograph and one MRI
  XR1MR3C1CnnTrf  C4  "XR + DESS/TSE/T2 + clinical"             = the same pattern with three MRI

They are built from the reference's own blocks only (trunks, FeaT, FeatC1) and follow its naming scheme
(`_fe{i}` / `_fe{i}_drop` by input position, `_agg_{i}` per MRI, `_agg_final`), so XR1MR2C1CnnTrf state dicts
load into the first 2 MRI slots of XR1MR3C1CnnTrf except `_fe3` (clinical there, MRI here) and `_agg_final`
(position embedding length).  Parity: the shared blocks are pinned by fixtures F2-F8; the hierarchical COMPOSITION
(`_HierFusionC1`: lane / token order, per-MRI aggregators without cls token, `_agg_final` sizing) is pinned to the reference
by running the generic class at its default (n_xr, n_mr) = (1, 2) on the reference's XR1MR2C1CnnTrf config against fixture F6
(tests/test_models_gpu.py::test_generic_hierarchy_reproduces_the_reference_class; the oracle's generic statement likewise,
tests/test_oracle_golden.py) -- XR1MR3C1CnnTrf / MR1C1CnnTrf are the same loop with another MRI count, checked against the
oracle's statement (tests/test_ext_gpu.py).  XR1C1Cnn's concatenation has no reference class to pin against."""
import math

import torch
from torch import nn

from .. import functional as KF
from . import _common as C
from ._xr1mrN import _feat, _shapes
from ._xrNmrMcP import FeatC1


class XR1C1Cnn(nn.Module):
    """forward(xr (B,1,R,C), clin (B,1,F))"""

    def __init__(self, config, path_weights):
        super().__init__()
        self.config = config
        fe = self.config["fe"]
        arch = fe["xr"]["arch"]
        if arch not in C.MAPPING_CH:
            raise ValueError("Unknown `num_elems` for `model.fe.xr` output")
        self._fe = C.build_trunk(arch, fe["xr"]["pretrained"], with_gap=True)
        self._fe_clin = FeatC1(config=fe["clin"])
        num_elems = C.MAPPING_CH[arch] + fe["clin"]["dim_out"]
        self.vs = {"fe_out_ch": C.MAPPING_CH[arch], "agg_in_len": num_elems}
        a = self.config["agg"]
        self._agg = nn.Sequential(nn.Dropout(a["dropout"]), nn.Linear(num_elems, a["hidden_size"]), nn.ReLU(),
                                  nn.Dropout(a["dropout"]))
        self._final = nn.Linear(a["hidden_size"], self.config["output_channels"])
        C.maybe_restore(self, self.config, path_weights)

    def forward(self, input0, input1):
        C.adopt(self, input0, input1)
        B = input0.shape[0]
        f_xr = self._fe(input0).reshape(B, -1)
        f_cl = self._fe_clin(input1).reshape(B, -1)
        t = KF.dropout(torch.cat([f_xr, f_cl], dim=1), self._agg[0].p, self.training)
        l1 = self._agg[1]
        t = KF.dropout(KF.relu(KF.linear(t, l1.weight, l1.bias)), self._agg[3].p, self.training)
        return C.finish(self.config, KF.linear(t, self._final.weight, self._final.bias))


class _HierFusionC1(nn.Module):
    """[XR] + n_mr MRI + clinical: per-MRI FeaT (no cls token) -> one FeaT over
    [XR tokens | MRI-1 tokens | ... | clinical token]."""
    n_xr, n_mr = 1, 2

    def __init__(self, config, path_weights):
        super().__init__()
        self.config = config
        if self.config["debug"]:
            print("Config at model init", self.config)
        fe, nx, nm = self.config["fe"], self.n_xr, self.n_mr
        n_in = nx + nm + 1
        self.i_clin = n_in - 1
        if len(self.config["input_size"]) != n_in or len(self.config["agg"]["num_slices"]) != n_in:
            raise ValueError(f"{type(self).__name__}: `input_size` and `agg.num_slices` need {n_in} entries")
        gap = bool((nx and fe["xr"]["with_gap"]) or fe["mr"]["with_gap"])
        self.vs = dict()
        shapes = _shapes(self.config, n_in)
        m = C.MAPPING_SPAT
        ns = self.config["agg"]["num_slices"]
        assert fe["mr"]["arch"] in C.MAPPING_CH
        d = C.MAPPING_CH[fe["mr"]["arch"]]
        for i in range(n_in):
            self.vs[f"fe{i}_shape_in"] = shapes[i]
            if i < nx:
                assert fe["xr"]["arch"] in C.MAPPING_CH and all(e in m for e in shapes[i])
                setattr(self, f"_fe{i}", C.build_trunk(fe["xr"]["arch"], fe["xr"]["pretrained"], gap))
                setattr(self, f"_fe{i}_drop", C.make_drop(fe["xr"]["dropout"]))
                self.vs[f"fe{i}_out_ch"] = C.MAPPING_CH[fe["xr"]["arch"]]
                self.vs[f"fe{i}_out_spat"] = (1, 1) if fe["xr"]["with_gap"] else tuple(m[e] for e in shapes[i])
                self.vs[f"agg_in_len_{i}"] = math.prod(self.vs[f"fe{i}_out_spat"])
            elif i < nx + nm:
                assert all(e in m for e in shapes[i][:2])
                setattr(self, f"_fe{i}", C.build_trunk(fe["mr"]["arch"], fe["mr"]["pretrained"], gap))
                setattr(self, f"_fe{i}_drop", C.make_drop(fe["mr"]["dropout"]))
                self.vs[f"fe{i}_out_ch"] = d
                self.vs[f"fe{i}_out_spat"] = (1, 1) if fe["mr"]["with_gap"] else tuple(m[e] for e in shapes[i][:2])
                self.vs[f"agg_in_len_{i}"] = ns[i] * math.prod(self.vs[f"fe{i}_out_spat"])
            else:
                setattr(self, f"_fe{i}", FeatC1(config=fe["clin"]))
                setattr(self, f"_fe{i}_drop", nn.Identity())
                self.vs[f"fe{i}_out_spat"] = (1, )
                self.vs[f"agg_in_len_{i}"] = ns[i]
        self.vs["agg_in_depth"] = d
        for i in range(nx, nx + nm):
            setattr(self, f"_agg_{i}", _feat(self.config, self.vs[f"agg_in_len_{i}"], d, with_cls=False))
        self._agg_final = _feat(self.config, sum(self.vs[f"agg_in_len_{i}"] for i in range(n_in)), d)
        C.maybe_restore(self, self.config, path_weights)

    def forward(self, *inputs):
        """inputs: n_xr radiographs (B,1,R,C), n_mr MRI volumes (B,1,R,C,S), clinical (B,1,F)"""
        if len(inputs) != self.i_clin + 1:
            raise TypeError(f"{type(self).__name__}.forward takes {self.i_clin + 1} inputs, got {len(inputs)}")
        C.adopt(self, *inputs)
        B = inputs[0].shape[0]
        nx, nm = self.n_xr, self.n_mr

        def agg(drop, feat):
            return lambda f: feat(C.tokens(drop(f), B))[1]
        # MRI trunks first (largest), each followed on its lane by its own aggregator; radiographs last
        jobs = [(getattr(self, f"_fe{i}"), inputs[i], C.mr_view(self.config),
                 agg(getattr(self, f"_fe{i}_drop"), getattr(self, f"_agg_{i}"))) for i in range(nx, nx + nm)]
        jobs += [(getattr(self, f"_fe{i}"), inputs[i], None) for i in range(nx)]
        res = C.run_trunks(jobs)
        toks = [C.tokens(getattr(self, f"_fe{i}_drop")(res[nm + i]), B) for i in range(nx)]
        toks += list(res[:nm])
        toks.append(getattr(self, f"_fe{self.i_clin}")(inputs[self.i_clin]))
        res_agg_final, _, _ = self._agg_final(torch.cat(toks, dim=1))
        return C.finish(self.config, res_agg_final.reshape(B, -1))


class MR1C1CnnTrf(_HierFusionC1):
    n_xr, n_mr = 0, 1


class XR1MR3C1CnnTrf(_HierFusionC1):
    n_xr, n_mr = 1, 3
