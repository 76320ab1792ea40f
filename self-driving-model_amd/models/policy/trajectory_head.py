"""EasyBackbone / TrajectoryPolicy -- drop-in for models/policy/trajectory_head.py:5-63."""
from typing import Dict, Optional

import torch
import torch.nn as nn

from ... import runtime
from ...hip import conv as hconv
from ...hip import ops as hops
from .._nn import BatchNorm2d, Conv2d, Linear, MLPSequential, ReLU, conv_bn_act, grouped_linear


class EasyBackbone(nn.Module):
    def __init__(self, in_channels: int = 3, out_dim: int = 512):
        super().__init__()
        if in_channels != 3:
            raise ValueError("EasyBackbone: the HIP first-layer kernel is built for 3-channel images")
        self.net = nn.Sequential(
            Conv2d(in_channels, 32, kernel_size=5, stride=2, padding=2), BatchNorm2d(32), nn.ReLU(inplace=True),
            Conv2d(32, 64, kernel_size=3, stride=2, padding=1), BatchNorm2d(64), nn.ReLU(inplace=True),
            Conv2d(64, 128, kernel_size=3, stride=2, padding=1), BatchNorm2d(128), nn.ReLU(inplace=True),
            Conv2d(128, 256, kernel_size=3, stride=2, padding=1), BatchNorm2d(256), nn.ReLU(inplace=True),
        )
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = Linear(256, out_dim)

    def forward(self, x: torch.Tensor, nhwc_input: Optional[torch.Tensor] = None) -> torch.Tensor:
        if nhwc_input is None:
            runtime.begin_step(x.device)
        h = nhwc_input if nhwc_input is not None else hops.image_to_nhwc(x, runtime.compute_dtype())
        for i in range(0, 12, 3):
            h = conv_bn_act(h, self.net[i], self.net[i + 1], relu=True)
        if nhwc_input is None:
            hconv.flush_bn_counters()
        return self.fc(hops.GapNhwc.apply(h, runtime.loss_scale()))


class TrajectoryPolicy(nn.Module):
    def __init__(self, horizon: int = 8, context_dim: int = 0, backbone_dim: int = 512):
        super().__init__()
        self.horizon = horizon
        self.backbone = EasyBackbone(in_channels=3, out_dim=backbone_dim)
        head_in_dim = backbone_dim + (context_dim if context_dim > 0 else 0)
        hidden = 512
        self.head_wp = MLPSequential(Linear(head_in_dim, hidden), ReLU(inplace=True), Linear(hidden, hidden), ReLU(inplace=True),
                                     Linear(hidden, horizon * 2))
        self.head_spd = MLPSequential(Linear(head_in_dim, hidden), ReLU(inplace=True), Linear(hidden, hidden), ReLU(inplace=True),
                                      Linear(hidden, horizon))
        self.group_heads = True  # tests flip this to compare with one launch per layer and head

    def forward(self, image: torch.Tensor, context: Optional[torch.Tensor] = None,
                nhwc_input: Optional[torch.Tensor] = None, backbone_feat: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        # backbone_feat: the backbone's output when the caller already ran it (AutoMoE overlaps it with the gating MLPs)
        feat = backbone_feat if backbone_feat is not None else self.backbone(image, nhwc_input=nhwc_input)
        x = torch.cat([feat, context], dim=1) if context is not None else feat
        if self.group_heads and x.is_cuda:
            # the two heads are independent 3-layer MLPs on the same input: each layer of both in one launch
            h = [x, x]
            for i in (0, 2, 4):
                h = grouped_linear([self.head_wp[i], self.head_spd[i]], h, relu=(i < 4))
            wp, spd = h
        else:
            wp, spd = self.head_wp(x), self.head_spd(x)
        return {"waypoints": wp.view(-1, self.horizon, 2), "speed": spd.view(-1, self.horizon)}
