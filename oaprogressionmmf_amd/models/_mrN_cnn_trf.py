"""MR1CnnTrf / MR2CnnTrf -- slice-wise CNN + transformer over slice tokens
(reference: koafusion/models/_mrN_cnn_trf.py:12-272)."""
import math

import torch
from torch import nn

from . import _common as C
from ._core_trf import FeaT


class MR1CnnTrf(nn.Module):
    def __init__(self, config, path_weights):
        super().__init__()
        self.config = config
        if self.config["debug"]:
            print("Config at model init", self.config)
        self.vs = dict()
        arch = self.config["fe"]["arch"]
        self._fe = C.build_trunk(arch, self.config["fe"]["pretrained"], self.config["fe"]["with_gap"])
        self._fe_drop = C.make_drop(self.config["fe"]["dropout"])
        if arch in ("resnet18", "resnet34"):
            self.vs["fe_out_ch"] = 512
        elif arch == "resnet50":
            self.vs["fe_out_ch"] = 2048
        else:
            raise ValueError("Unsupported `model.fe.arch`")
        t = self.config["input_size"][0]
        if self.config["downscale"]:
            t = [round(s * d) for s, d in zip(t, self.config["downscale"][0])]
        self.vs["shape_in"] = t
        if self.config["fe"]["with_gap"]:
            self.vs["fe_out_spat"] = (1, 1, 1)
        else:
            try:
                mapping = {320: 10, 160: 5, 128: 4, 96: 3, 64: 2, 32: 1}
                self.vs["fe_out_spat"] = tuple(mapping[e] for e in self.vs["shape_in"])
            except (ValueError, IndexError):
                raise ValueError("Unspecified `model.fe` output shape for given `model.input_size`")
        sp = self.vs["fe_out_spat"]
        dv = self.config["fe"]["dims_view"]
        if dv == "rc":
            self.vs["agg_in_len"] = self.vs["shape_in"][2] * (sp[0] * sp[1])
        elif dv == "cs":
            self.vs["agg_in_len"] = self.vs["shape_in"][0] * (sp[1] * sp[2])
        elif dv == "rs":
            self.vs["agg_in_len"] = self.vs["shape_in"][1] * (sp[0] * sp[2])
        else:
            raise ValueError("Unsupported `model.fe.dims_view`")
        self.vs["agg_in_depth"] = self.vs["fe_out_ch"]
        a = self.config["agg"]
        self._agg = FeaT(num_patches=self.vs["agg_in_len"], patch_dim=self.vs["agg_in_depth"],
                         emb_dim=self.vs["agg_in_depth"], depth=a["depth"], heads=a["heads"], mlp_dim=a["mlp_dim"],
                         num_classes=self.config["output_channels"], emb_dropout=a["emb_dropout"],
                         mlp_dropout=a["mlp_dropout"])
        C.maybe_restore(self, self.config, path_weights)

    def forward(self, input):
        """input : (B, 1, R, C, S)"""
        C.adopt(self, input)
        B = input.shape[0]
        t_in = C.fold_slices(input, self.config["fe"]["dims_view"])
        t_fe = self._fe_drop(self._fe(t_in))
        t_fe = C.tokens(t_fe, B)
        res_agg, _, _ = self._agg(t_fe)
        return C.finish(self.config, res_agg.reshape(B, -1))


class MR2CnnTrf(nn.Module):
    def __init__(self, config, path_weights):
        super().__init__()
        self.config = config
        if self.config["debug"]:
            print("Config at model init", self.config)
        self.vs = dict()
        arch = self.config["fe"]["arch"]
        self._fe0 = C.build_trunk(arch, self.config["fe"]["pretrained"], self.config["fe"]["with_gap"])
        self._fe1 = C.build_trunk(arch, self.config["fe"]["pretrained"], self.config["fe"]["with_gap"])
        self._fe0_drop = C.make_drop(self.config["fe"]["dropout"])
        self._fe1_drop = C.make_drop(self.config["fe"]["dropout"])
        if arch in ("resnet18", "resnet34"):
            self.vs["fe_out_ch"] = 512
        elif arch == "resnet50":
            self.vs["fe_out_ch"] = 2048
        else:
            raise ValueError("Unsupported `model.fe.arch`")
        if self.config["fe"]["with_gap"]:
            self.vs["fe_out_spat"] = (1, 1)
        else:
            if self.config["input_size"][0][0] == 320:
                self.vs["fe_out_spat"] = (5, 5)
            else:
                raise ValueError("Unspecified `model.fe` output shape for given `model.input_size`")
        a = self.config["agg"]
        self.vs["agg_in_len"] = (a["num_slices"][0] + a["num_slices"][1]) * math.prod(self.vs["fe_out_spat"])
        self.vs["agg_in_depth"] = self.vs["fe_out_ch"]
        self._agg = FeaT(num_patches=self.vs["agg_in_len"], patch_dim=self.vs["agg_in_depth"],
                         emb_dim=self.vs["agg_in_depth"], depth=a["depth"], heads=a["heads"], mlp_dim=a["mlp_dim"],
                         num_classes=self.config["output_channels"], emb_dropout=a["emb_dropout"],
                         mlp_dropout=a["mlp_dropout"])
        C.maybe_restore(self, self.config, path_weights)

    def forward(self, input0, input1):
        """input0, input1 : (B, 1, R, C, S)"""
        C.adopt(self, input0, input1)
        B = input0.shape[0]
        f0, f1 = C.run_trunks([(self._fe0, input0, C.mr_view(self.config)), (self._fe1, input1, C.mr_view(self.config))])
        t_fe0 = C.tokens(self._fe0_drop(f0), B)
        t_fe1 = C.tokens(self._fe1_drop(f1), B)
        t_fe_m = torch.cat([t_fe0, t_fe1], dim=1)
        res_agg, _, _ = self._agg(t_fe_m)
        return C.finish(self.config, res_agg.reshape(B, -1))
