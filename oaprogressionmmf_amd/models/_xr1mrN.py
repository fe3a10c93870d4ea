"""XR1MR1CnnTrf (flat fusion) / XR1MR2CnnTrf (hierarchical fusion)
(reference: koafusion/models/_xr1mrN.py:11-369)."""
import math

import torch
from torch import nn

from . import _common as C
from ._core_trf import FeaT


def _feat(cfg, n, depth_ch, with_cls=True):
    a = cfg["agg"]
    return FeaT(num_patches=n, patch_dim=depth_ch, emb_dim=depth_ch, depth=a["depth"], heads=a["heads"],
                mlp_dim=a["mlp_dim"], num_classes=cfg["output_channels"], emb_dropout=a["emb_dropout"],
                with_cls=with_cls, mlp_dropout=a["mlp_dropout"])


def _shapes(cfg, n):
    out = []
    for i in range(n):
        t = cfg["input_size"][i]
        if cfg["downscale"]:
            t = [round(s * d) for s, d in zip(t, cfg["downscale"][i])]
        out.append(t)
    return out


class XR1MR1CnnTrf(nn.Module):
    def __init__(self, config, path_weights):
        super().__init__()
        self.config = config
        if self.config["debug"]:
            print("Config at model init", self.config)
        self.vs = dict()
        fe = self.config["fe"]
        gap = bool(fe["xr"]["with_gap"] or fe["mr"]["with_gap"])   # Q7: one flag decides for all trunks
        self._fe0 = C.build_trunk(fe["xr"]["arch"], fe["xr"]["pretrained"], gap)
        self._fe1 = C.build_trunk(fe["mr"]["arch"], fe["mr"]["pretrained"], gap)
        self._fe0_drop = C.make_drop(fe["xr"]["dropout"])
        self._fe1_drop = C.make_drop(fe["mr"]["dropout"])
        assert fe["xr"]["arch"] in C.MAPPING_CH
        assert fe["mr"]["arch"] in C.MAPPING_CH
        self.vs["fe0_out_ch"] = C.MAPPING_CH[fe["xr"]["arch"]]
        self.vs["fe1_out_ch"] = C.MAPPING_CH[fe["mr"]["arch"]]
        t_0, t_1 = _shapes(self.config, 2)
        self.vs["fe0_shape_in"], self.vs["fe1_shape_in"] = t_0, t_1
        m = C.MAPPING_SPAT
        assert all(e in m for e in t_0)
        assert all(e in m for e in t_1[:2])
        self.vs["fe0_out_spat"] = (1, 1) if fe["xr"]["with_gap"] else tuple(m[e] for e in t_0)
        self.vs["fe1_out_spat"] = (1, 1) if fe["mr"]["with_gap"] else tuple(m[e] for e in t_1[:2])
        self.vs["agg_in_len_0"] = math.prod(self.vs["fe0_out_spat"])
        self.vs["agg_in_len_1"] = self.config["agg"]["num_slices"][1] * math.prod(self.vs["fe1_out_spat"])
        self.vs["agg_in_depth"] = self.vs["fe1_out_ch"]
        self._agg = _feat(self.config, self.vs["agg_in_len_0"] + self.vs["agg_in_len_1"], self.vs["agg_in_depth"])
        C.maybe_restore(self, self.config, path_weights)

    def forward(self, input0, input1):
        """input0 : (B,1,R,C); input1 : (B,1,R,C,S)"""
        C.adopt(self, input0, input1)
        B = input0.shape[0]
        f1, f0 = C.run_trunks([(self._fe1, input1, C.mr_view(self.config)), (self._fe0, input0, None)])
        t_fe0 = C.tokens(self._fe0_drop(f0), B)
        t_fe1 = C.tokens(self._fe1_drop(f1), B)
        res_agg, _, _ = self._agg(torch.cat([t_fe0, t_fe1], dim=1))
        return C.finish(self.config, res_agg.reshape(B, -1))


class XR1MR2CnnTrf(nn.Module):
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
        self._fe0_drop = C.make_drop(fe["xr"]["dropout"])
        self._fe1_drop = C.make_drop(fe["mr"]["dropout"])
        self._fe2_drop = C.make_drop(fe["mr"]["dropout"])
        assert fe["xr"]["arch"] in C.MAPPING_CH
        assert fe["mr"]["arch"] in C.MAPPING_CH
        self.vs["fe0_out_ch"] = C.MAPPING_CH[fe["xr"]["arch"]]
        self.vs["fe12_out_ch"] = C.MAPPING_CH[fe["mr"]["arch"]]
        t_0, t_1, t_2 = _shapes(self.config, 3)
        self.vs["fe0_shape_in"], self.vs["fe1_shape_in"], self.vs["fe2_shape_in"] = t_0, t_1, t_2
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
        ns = self.config["agg"]["num_slices"]
        self.vs["agg_in_len_0"] = math.prod(self.vs["fe0_out_spat"])
        self.vs["agg_in_len_1"] = ns[1] * math.prod(self.vs["fe1_out_spat"])
        self.vs["agg_in_len_2"] = ns[2] * math.prod(self.vs["fe2_out_spat"])
        self.vs["agg_in_depth"] = self.vs["fe12_out_ch"]
        d = self.vs["agg_in_depth"]
        self._agg_1 = _feat(self.config, self.vs["agg_in_len_1"], d, with_cls=False)
        self._agg_2 = _feat(self.config, self.vs["agg_in_len_2"], d, with_cls=False)
        self._agg_final = _feat(self.config, self.vs["agg_in_len_0"] + self.vs["agg_in_len_1"] +
                                self.vs["agg_in_len_2"], d)
        C.maybe_restore(self, self.config, path_weights)

    def forward(self, input0, input1, input2):
        C.adopt(self, input0, input1, input2)
        B = input0.shape[0]
        # each MRI's aggregator runs on its encoder's lane (Q4: its mlp_head0 output is computed and discarded)
        def agg(drop, feat):
            return lambda f: feat(C.tokens(drop(f), B))[1]
        res_agg1, res_agg2, f0 = C.run_trunks([(self._fe1, input1, C.mr_view(self.config), agg(self._fe1_drop, self._agg_1)),
                                               (self._fe2, input2, C.mr_view(self.config), agg(self._fe2_drop, self._agg_2)),
                                               (self._fe0, input0, None)])
        t_fe0 = C.tokens(self._fe0_drop(f0), B)
        # Q4: the reference runs mlp_head0 of the cls-less aggregators and discards it
        res_agg_final, _, _ = self._agg_final(torch.cat([t_fe0, res_agg1, res_agg2], dim=1))
        return C.finish(self.config, res_agg_final.reshape(B, -1))
