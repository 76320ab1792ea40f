"""ResNet-18 trunk (torchvision `resnet18`, children()[:-2]) as the reference uses it at
models/experts/bdd_detection_expert.py:9-10, bdd_segmentation_expert.py:10-11, bdd_drivable_expert.py:10-11.

Same child order and names as torchvision, so the state_dict keys are `backbone.0.weight`,
`backbone.1.running_mean`, `backbone.4.0.conv1.weight`, `backbone.5.0.downsample.0.weight`, ...
Compute: NHWC gather-GEMM convs with BatchNorm batch statistics in the conv epilogue and
normalise + residual + ReLU in one elementwise pass (hip/conv.py).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ... import runtime
from ...hip import conv as hconv
from ...hip import ops as hops
from .._nn import BatchNorm2d, Conv2d, conv_bn_act


FUSE_STEM_POOL = os.environ.get("AUTOMOE_FUSE_STEM_POOL", "1") != "0"  # tests flip this to compare with conv -> BN -> ReLU -> separate max-pool


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(Conv2d(cin, cout, 1, stride, 0, bias=False), BatchNorm2d(cout))
        self.stride = stride

    def forward(self, x):  # x: NHWC
        if isinstance(x, hconv.PendingAffine):
            # the pooled stem with its BatchNorm + ReLU still pending (one-pass frozen stem): the fused frozen block forms it in
            # conv1's input staging and in the block-end pass; any other case writes the normalised map first
            y = None
            if self.downsample is None and self.conv1.in_channels == 64 and not self.conv1.weight.requires_grad:
                y = hconv.fused_basic_block_identity(x.raw, self.conv1.weight, self.bn1, self.conv1._packed, self.conv2.weight, self.bn2,
                                                     self.conv2._packed, pre=(x.scale, x.shift))
            if y is not None:
                return y
            x = x.materialize()
        if self.downsample is None and self.conv1.in_channels in (64, 128) and (not torch.is_grad_enabled() or not self.conv1.weight.requires_grad):
            y = hconv.fused_basic_block_identity(x, self.conv1.weight, self.bn1, self.conv1._packed, self.conv2.weight, self.bn2,
                                                 self.conv2._packed)  # frozen layer1 / layer2 block: bn1 + ReLU live inside conv2's input staging
            if y is not None:
                return y
        if self.downsample is not None and not self.conv1.weight.requires_grad:
            y = hconv.fused_basic_block_down(x, self)  # frozen strided block: the shortcut's BN rides in the final add + ReLU pass
            if y is not None:
                return y
        idn = x if self.downsample is None else conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        # identity shortcut: the block input gets two gradients (through conv1 and through the shortcut); the block end hands the
        # shortcut's to conv1, whose input-gradient kernel adds it in its epilogue (torchvision BasicBlock.forward `out += identity`
        # in backward: no separate accumulate pass over the block input)
        merge = self.downsample is None and torch.is_grad_enabled() and x.requires_grad
        y = conv_bn_act(x, self.conv1, self.bn1, relu=True, take_residual_grad=merge)
        return conv_bn_act(y, self.conv2, self.bn2, relu=True, residual=idn, give_residual_grad=merge)


class MaxPool(nn.MaxPool2d):
    def __init__(self):
        super().__init__(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        return hops.MaxPool3x3s2.apply(x)


class Trunk(nn.Sequential):
    """Sequential(conv1, bn1, relu, maxpool, layer1, layer2, layer3, layer4); forward takes and returns NHWC."""

    def __init__(self):
        def stage(cin, cout, stride):
            return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))

        super().__init__(Conv2d(3, 64, 7, 2, 3, bias=False), BatchNorm2d(64), nn.ReLU(inplace=True), MaxPool(),
                         stage(64, 64, 1), stage(64, 128, 2), stage(128, 256, 2), stage(256, 512, 2))
        for m in self.modules():  # torchvision ResNet.__init__
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def forward(self, x):
        pooled = None
        if not torch.is_grad_enabled() or not any(p.requires_grad for p in (self[0].weight, self[1].weight, self[1].bias)):
            cfg = hconv._Cfg(self[0].spec, self[0]._packed, self[1], True, runtime.loss_scale(), getattr(x, "orig_hw", None))
            # frozen stem: conv + BN + ReLU + maxpool in one light pass (train-mode BatchNorm: a PendingAffine for layer1's first
            # block) or two
            pooled = hconv.fused_stem_pool(x, self[0].weight, self[1], cfg, allow_pending=True)
        if pooled is None:
            # trainable stem: normalise + ReLU + max-pool as one pass over the raw conv output when the fused pass applies (the
            # call then returns the pooled map); otherwise the separate pool
            full_hw = getattr(x, "orig_hw", None)
            y = conv_bn_act(x, self[0], self[1], relu=True, pool=FUSE_STEM_POOL)
            conv_hw = None if full_hw is None else (hconv.out_size(full_hw[0], self[0].spec), hconv.out_size(full_hw[1], self[0].spec))
            x = y if (conv_hw is not None and tuple(y.shape[1:3]) != conv_hw) else self[3](y)
        else:
            x = pooled
        for i in range(4, 8):
            for blk in self[i]:
                x = blk(x)
        return x


def load_pretrained_(trunk: Trunk, flag: bool):
    """pretrained_backbone=True means torchvision's ImageNet weights: a network fetch in the reference.
    Offline it needs a local state_dict: set AUTOMOE_RESNET18_WEIGHTS to a torchvision resnet18 .pth file."""
    if not flag:
        return
    import os
    path = os.environ.get("AUTOMOE_RESNET18_WEIGHTS", "")
    if not path or not os.path.exists(path):
        raise RuntimeError("pretrained_backbone=True needs torchvision's ImageNet ResNet-18 weights, which the reference "
                           "downloads; there is no network here. Point AUTOMOE_RESNET18_WEIGHTS at a local resnet18 "
                           "state_dict (.pth) or pass pretrained_backbone=False.")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    names = ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4"]
    remap = {}
    for k, v in sd.items():
        head = k.split(".")[0]
        if head in names:
            remap[str(names.index(head)) + k[len(head):]] = v
    trunk.load_state_dict(remap, strict=True)
