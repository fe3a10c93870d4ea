"""XR1Cnn -- single-radiograph CNN (reference: koafusion/models/_xr1_cnn.py:9-81)."""
from torch import nn

from .. import functional as KF
from . import _common as C


class XR1Cnn(nn.Module):
    def __init__(self, config, path_weights):
        super().__init__()
        self.config = config
        if self.config["debug"]:
            print("Config at model init", self.config)
        arch = self.config["fe"]["arch"]
        self._fe = C.build_trunk(arch, self.config["fe"]["pretrained"], with_gap=True)
        if arch in ("resnet18", "resnet34"):
            num_elems = 512
        elif arch in ("resnet50", "resnext50_32x4d"):
            num_elems = 2048
        else:
            raise ValueError("Unknown `num_elems` for `model.fe` output. Get via `model.debug=true`")
        self._agg = nn.Sequential(
            nn.Dropout(self.config["agg"]["dropout"]),
            nn.Linear(num_elems, self.config["agg"]["hidden_size"]),
            nn.ReLU(),
            nn.Dropout(self.config["agg"]["dropout"]),
        )
        self._final = nn.Linear(self.config["agg"]["hidden_size"], self.config["output_channels"])
        C.maybe_restore(self, self.config, path_weights)

    def forward(self, input):
        """input : (B, 1, R, C)"""
        C.adopt(self, input)
        res_fe = self._fe(input)                       # (B, F, 1, 1); the k=3 channel repeat is folded
        tmp_fe = res_fe.reshape(res_fe.shape[0], -1)   # "b ch d0 d1 -> b (ch d0 d1)"
        p = self._agg[0].p
        t = KF.dropout(tmp_fe, p, self.training)
        l1 = self._agg[1]
        t = KF.relu(KF.linear(t, l1.weight, l1.bias))
        t = KF.dropout(t, self._agg[3].p, self.training)
        res_out = KF.linear(t, self._final.weight, self._final.bias)
        return C.finish(self.config, res_out)
