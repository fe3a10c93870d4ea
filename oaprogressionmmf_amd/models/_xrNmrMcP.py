"""XR1MR2C1CnnTrf -- radiograph + 2 MRI volumes + clinical vector, hierarchical transformer fusion
(reference: koafusion/models/_xrNmrMcP.py:11-264)."""
import math

import torch
from torch import nn

from .. import functional as KF
from . import _common as C
from ._xr1mrN import _feat, _shapes


class FeatC1(nn.Module):
    """Linear(dim_in, dim_out) -> GELU -> Dropout on the clinical vector (_xrNmrMcP.py:11-29)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self._fe = nn.Sequential(
            nn.Linear(self.config["dim_in"], self.config["dim_out"]),
            nn.GELU(),
            nn.Dropout(self.config["dropout"]),
        )

    def forward(self, input_):
        l0 = self._fe[0]
        t = KF.gelu(KF.linear(input_, l0.weight, l0.bias))
        return KF.dropout(t, self._fe[2].p, self.training)


class XR1MR2C1CnnTrf(nn.Module):
    def __init__(self, config, path_weights):
        super().__init__()
        self.config = config
        if self.config["debug"]:
            print("Config at model init", self.config)
        self.vs = dict()
        fe = self.config["fe"]
        gap = bool(fe["xr"]["with_gap"] or fe["mr"]["with_gap"])
        self._fe0 = C.build_trunk(fe["xr"]["arch"], fe["xr"]["pretrained"], gap)
        self._fe1 = C.build_trunk(fe["mr"]["arch"], fe["mr"]["pretrained"], gap)
        self._fe2 = C.build_trunk(fe["mr"]["arch"], fe["mr"]["pretrained"], gap)
        self._fe3 = FeatC1(config=fe["clin"])
        self._fe0_drop = C.make_drop(fe["xr"]["dropout"])
        self._fe1_drop = C.make_drop(fe["mr"]["dropout"])
        self._fe2_drop = C.make_drop(fe["mr"]["dropout"])
        self._fe3_drop = nn.Identity()
        assert fe["xr"]["arch"] in C.MAPPING_CH
        assert fe["mr"]["arch"] in C.MAPPING_CH
        self.vs["fe0_out_ch"] = C.MAPPING_CH[fe["xr"]["arch"]]
        self.vs["fe12_out_ch"] = C.MAPPING_CH[fe["mr"]["arch"]]
        t_0, t_1, t_2, t_3 = _shapes(self.config, 4)
        self.vs["fe0_shape_in"], self.vs["fe1_shape_in"] = t_0, t_1
        self.vs["fe2_shape_in"], self.vs["fe3_shape_in"] = t_2, t_3
        m = C.MAPPING_SPAT
        assert all(e in m for e in t_0)
        assert all(e in m for e in t_1[:2])
        assert all(e in m for e in t_2[:2])
        self.vs["fe0_out_spat"] = (1, 1) if fe["xr"]["with_gap"] else tuple(m[e] for e in t_0)
        if fe["mr"]["with_gap"]:
            self.vs["fe1_out_spat"] = (1, 1)
            self.vs["fe2_out_spat"] = (1, 1)
        else:
            self.vs["fe1_out_spat"] = tuple(m[e] for e in t_1[:2])
            self.vs["fe2_out_spat"] = tuple(m[e] for e in t_2[:2])
        self.vs["fe3_out_spat"] = (1, )
        ns = self.config["agg"]["num_slices"]
        self.vs["agg_in_len_0"] = math.prod(self.vs["fe0_out_spat"])
        self.vs["agg_in_len_1"] = ns[1] * math.prod(self.vs["fe1_out_spat"])
        self.vs["agg_in_len_2"] = ns[2] * math.prod(self.vs["fe2_out_spat"])
        self.vs["agg_in_len_3"] = ns[3] * math.prod(self.vs["fe3_out_spat"])
        self.vs["agg_in_depth"] = self.vs["fe12_out_ch"]
        d = self.vs["agg_in_depth"]
        self._agg_1 = _feat(self.config, self.vs["agg_in_len_1"], d, with_cls=False)
        self._agg_2 = _feat(self.config, self.vs["agg_in_len_2"], d, with_cls=False)
        self._agg_final = _feat(self.config, self.vs["agg_in_len_0"] + self.vs["agg_in_len_1"] +
                                self.vs["agg_in_len_2"] + self.vs["agg_in_len_3"], d)
        C.maybe_restore(self, self.config, path_weights)

    def forward(self, input0, input1, input2, input3):
        """input0 (B,1,R,C) radiograph; input1, input2 (B,1,R,C,S) MRI volumes; input3 (B,1,F) clinical"""
        C.adopt(self, input0, input1, input2, input3)
        B = input0.shape[0]
        # largest encoder first so the short ones fill its tails
        # each MRI's aggregator runs on its encoder's lane (Q4: its mlp_head0 output is computed and discarded)
        def agg(drop, feat):
            return lambda f: feat(C.tokens(drop(f), B))[1]
        res_agg1, res_agg2, f0 = C.run_trunks([(self._fe1, input1, C.mr_view(self.config), agg(self._fe1_drop, self._agg_1)),
                                               (self._fe2, input2, C.mr_view(self.config), agg(self._fe2_drop, self._agg_2)),
                                               (self._fe0, input0, None)])
        t_fe0 = C.tokens(self._fe0_drop(f0), B)
        t_fe3 = self._fe3_drop(self._fe3(input3))
        t_fe_m = torch.cat([t_fe0, res_agg1, res_agg2, t_fe3], dim=1)
        res_agg_final, _, _ = self._agg_final(t_fe_m)
        return C.finish(self.config, res_agg_final.reshape(B, -1))
