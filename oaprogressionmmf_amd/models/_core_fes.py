"""Encoder registry `dict_fes` (reference: koafusion/models/_core_fes.py:6-15) and the ResNet / ResNeXt
parameter containers (reference: koafusion/models/_torchvision.py:34-246, and torchvision's identical
resnet18/34/50 used through `torchvision.models`).

The classes below only OWN parameters/buffers under the reference's names (conv1, bn1, layer1.0.conv1,
layer1.0.downsample.0, ...), so `state_dict()` keys and shapes are identical to the reference and its
checkpoints load unchanged.  They are never executed layer by layer: the whole trunk runs through the
fused HIP schedule in `_encoder.py`.
"""
import math

import torch
from torch import nn


class Bottleneck(nn.Module):
    """Parameter container for a ResNet v1.5 bottleneck (stride on the 3x3; _torchvision.py:83-138)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("koaf Bottleneck is a parameter container; run the enclosing KoafTrunk")


class BasicBlock(nn.Module):
    """Parameter container for the resnet18/34 block (_torchvision.py:34-80)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("koaf BasicBlock is a parameter container; run the enclosing KoafTrunk")


class ResNet(nn.Module):
    """Children in the reference order: conv1, bn1, relu, maxpool, layer1..4, avgpool, fc -- the model
    classes slice `list(children())[:-1]` / `[:-2]` exactly like koafusion/models/_xrNmrMcP.py:47-56."""

    def __init__(self, block, layers, num_classes=1000, groups=1, width_per_group=64):
        super().__init__()
        self.inplanes = 64
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        # same initialisation law as _torchvision.py:185-190
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width))
        return nn.Sequential(*layers)

    def forward(self, x):
        raise RuntimeError("koaf ResNet is a parameter container; build a KoafTrunk from its children")


# the ImageNet checkpoints the reference downloads (koafusion/models/_torchvision.py:12-20; torchvision's own resnet18/34/50 with
# `pretrained=True`, which the MRI trunks go through, fetch the same files)
model_urls = {
    "resnet18": "https://download.pytorch.org/models/resnet18-f37072fd.pth",
    "resnet34": "https://download.pytorch.org/models/resnet34-b627a593.pth",
    "resnet50": "https://download.pytorch.org/models/resnet50-0676ba61.pth",
    "resnext50_32x4d": "https://download.pytorch.org/models/resnext50_32x4d-7cdf4587.pth",
}


def pretrained_candidates(arch):
    """where `pretrained=True` looks for the checkpoint of `arch`, in order: $KOAF_PRETRAINED_DIR/<file>, then the file
    torch.hub would have cached for the reference's `load_state_dict_from_url(model_urls[arch])`
    (<torch.hub.get_dir()>/checkpoints/<file>, i.e. $TORCH_HOME/hub/checkpoints)"""
    import os
    from pathlib import Path
    fname = model_urls[arch].rsplit("/", 1)[-1]
    out = []
    if os.environ.get("KOAF_PRETRAINED_DIR"):
        out.append(Path(os.environ["KOAF_PRETRAINED_DIR"]) / fname)
    out.append(Path(torch.hub.get_dir()) / "checkpoints" / fname)
    return out


def _load_pretrained(model, arch):
    """`pretrained=True` (koafusion/models/_torchvision.py:249-261: load_state_dict_from_url + load_state_dict) without a network:
    the checkpoint file is taken from the local torch-hub cache or $KOAF_PRETRAINED_DIR; the keys (fc included) are the
    reference's, so it loads strictly.  Absent file: RuntimeError naming the paths -- nothing is downloaded."""
    for path in pretrained_candidates(arch):
        if path.is_file():
            model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
            return model
    raise RuntimeError(
        f"{arch}(pretrained=True): the reference downloads {model_urls[arch]} (koafusion/models/_torchvision.py:258-261); there is "
        f"no network here and none of {[str(p) for p in pretrained_candidates(arch)]} exists -- place the file there (torch-hub cache "
        f"or $KOAF_PRETRAINED_DIR), or construct with pretrained=False and load a state_dict")


def _resnet(arch, block, layers, pretrained, **kw):
    model = ResNet(block, layers, **kw)
    return _load_pretrained(model, arch) if pretrained else model


def resnet18(pretrained=False, progress=True, **kw):
    return _resnet("resnet18", BasicBlock, [2, 2, 2, 2], pretrained, **kw)


def resnet34(pretrained=False, progress=True, **kw):
    return _resnet("resnet34", BasicBlock, [3, 4, 6, 3], pretrained, **kw)


def resnet50(pretrained=False, progress=True, **kw):
    return _resnet("resnet50", Bottleneck, [3, 4, 6, 3], pretrained, **kw)


def resnext50_32x4d(pretrained=False, progress=True, **kw):
    kw["groups"] = 32
    kw["width_per_group"] = 4
    return _resnet("resnext50_32x4d", Bottleneck, [3, 4, 6, 3], pretrained, **kw)


def _unsupported(name):
    def ctor(pretrained=False, **kw):
        raise NotImplementedError(
            f"encoder '{name}' is registered by the reference (koafusion/models/_core_fes.py:7-10) but no model "
            f"class can use it (every class maps arch -> channel width for resnet*/resnext50 only); not built")
    return ctor


# same keys as koafusion/models/_core_fes.py:6-15
dict_fes = {
    "squeezenet1_0": _unsupported("squeezenet1_0"),
    "vgg16": _unsupported("vgg16"),
    "densenet161": _unsupported("densenet161"),
    "inception_v3": _unsupported("inception_v3"),
    "resnet18": resnet18,
    "resnet34": resnet34,
    "resnet50": resnet50,
    "resnext50_32x4d": resnext50_32x4d,
}
