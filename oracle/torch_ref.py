"""torch-CPU fp32 restatement of the reference modules on the AutoMoE hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every class cites the reference lines it
restates; state_dict keys are identical to the reference's so one set of weights can be loaded
into the oracle, the reference (where importable) and the HIP product modules.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# ResNet-18 trunk as used at models/experts/bdd_detection_expert.py:9-10 (torchvision
# `resnet18`, children()[:-2]).  torchvision is absent from the image: restated from the
# published architecture (He et al. 2015; BasicBlock [2,2,2,2]) -- "parity unpinned",
# checked structurally: 11,176,512 parameters, key names as in SURVEY.md section 8(b).
# --------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idn)


def resnet18_trunk() -> nn.Sequential:
    """Sequential(conv1, bn1, relu, maxpool, layer1..layer4): indices 0..7 like the reference."""
    def stage(cin, cout, stride):
        return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))

    trunk = nn.Sequential(
        nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
        nn.MaxPool2d(3, 2, 1),
        stage(64, 64, 1), stage(64, 128, 2), stage(128, 256, 2), stage(256, 512, 2),
    )
    for m in trunk.modules():  # torchvision's ResNet.__init__ initialisation
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1.0)
            nn.init.constant_(m.bias, 0.0)
    return trunk


def _no_pretrained(flag: bool):
    if flag:
        raise RuntimeError("pretrained_backbone=True needs a network fetch (torchvision ImageNet weights); "
                           "unavailable offline -- pass pretrained_backbone=False")


class BDDDetectionExpert(nn.Module):
    """models/experts/bdd_detection_expert.py:4-31."""

    def __init__(self, num_classes: int = 10, pretrained_backbone: bool = True):
        super().__init__()
        _no_pretrained(pretrained_backbone)
        self.num_classes = num_classes
        self.backbone = resnet18_trunk()
        self.head = nn.Sequential(nn.Conv2d(512, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, num_classes + 4, 1))

    def forward(self, x):
        out = self.head(self.backbone(x))
        return {"class_logits": out[:, : self.num_classes], "bbox_deltas": out[:, self.num_classes:]}

    def predict(self, x):
        o = self.forward(x)
        return {"class_probs": o["class_logits"].softmax(dim=1), "bbox_deltas": o["bbox_deltas"].sigmoid()}


class _DenseExpert(nn.Module):
    """models/experts/bdd_segmentation_expert.py:5-23 and bdd_drivable_expert.py:5-23 (same body)."""

    def __init__(self, num_classes: int, pretrained_backbone: bool):
        super().__init__()
        _no_pretrained(pretrained_backbone)
        self.num_classes = num_classes
        self.backbone = resnet18_trunk()
        self.decoder = nn.Sequential(nn.Conv2d(512, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, num_classes, 1))

    def forward(self, x):
        low = self.decoder(self.backbone(x))
        return F.interpolate(low, size=x.shape[-2:], mode="bilinear", align_corners=False)


class BDDSegmentationExpert(_DenseExpert):
    def __init__(self, num_classes: int = 19, pretrained_backbone: bool = True):
        super().__init__(num_classes, pretrained_backbone)


class BDDDrivableExpert(_DenseExpert):
    def __init__(self, num_classes: int = 3, pretrained_backbone: bool = True):
        super().__init__(num_classes, pretrained_backbone)


class NuScenesExpert(nn.Module):
    """models/experts/nuscenes_expert.py:96-190, image branch (use_lidar=False, the AutoMoE configuration): resnet18
    children()[:-1] (trunk + AdaptiveAvgPool2d(1)) -> Linear(512,256) -> + query embeddings -> decoder -> heads."""

    def __init__(self, image_backbone=None, lidar_backbone=None, fusion: str = "concat", num_queries: int = 100,
                 use_lidar: bool = False, use_tnet: bool = False, bbox_dim: int = 7, pretrained_backbone: bool = True):
        super().__init__()
        assert not use_lidar and lidar_backbone is None, "LiDAR branch not restated"
        if image_backbone is None:
            _no_pretrained(pretrained_backbone)
            trunk = resnet18_trunk()
            trunk.append(nn.AdaptiveAvgPool2d((1, 1)))
            self.image_backbone = trunk
            self.image_projection = nn.Linear(512, 256)
        else:
            self.image_backbone = image_backbone
            self.image_projection = nn.Identity()
        self.num_queries, self.bbox_dim = num_queries, bbox_dim
        self.query_embed = nn.Embedding(num_queries, 256)
        self.decoder = nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Dropout(0.3), nn.Linear(256, 128), nn.ReLU(),
                                     nn.Dropout(0.3))
        self.class_head = nn.Linear(128, 10)
        self.bbox_head = nn.Linear(128, bbox_dim)

    def forward(self, batch):
        f = self.image_backbone(batch["image"])
        f = self.image_projection(f.view(f.size(0), -1))
        B = f.size(0)
        x = f.unsqueeze(1).expand(B, self.num_queries, -1) + self.query_embed.weight.unsqueeze(0).expand(B, -1, -1)
        x = self.decoder(x)
        return {"class_logits": self.class_head(x), "bbox_preds": self.bbox_head(x)}


# --------------------------------------------------------------------------------------------
# Extractors: models/experts/expert_extractors.py:20-137, 140-200
# --------------------------------------------------------------------------------------------
def _extractor_mlp(cin: int, out_dim: int) -> nn.Sequential:
    # indices 0 pool, 1 flatten, 2 linear, 3 relu, 4 dropout, 5 linear, 6 layernorm
    return nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(), nn.Linear(cin, 512), nn.ReLU(),
                         nn.Dropout(0.1), nn.Linear(512, out_dim), nn.LayerNorm(out_dim))


class DetectionExpertExtractor(nn.Module):
    def __init__(self, output_dim: int = 256, num_classes: int = 10):
        super().__init__()
        self.output_dim, self.num_classes = output_dim, num_classes
        self.feature_extractor = _extractor_mlp(num_classes + 4, output_dim)

    def forward(self, o):
        return self.feature_extractor(torch.cat([o["class_logits"], o["bbox_deltas"]], dim=1))


class SegmentationExpertExtractor(nn.Module):
    def __init__(self, output_dim: int = 256, num_classes: int = 19):
        super().__init__()
        self.output_dim, self.num_classes = output_dim, num_classes
        self.feature_extractor = _extractor_mlp(num_classes, output_dim)

    def forward(self, o):
        return self.feature_extractor(o)


class DrivableExpertExtractor(SegmentationExpertExtractor):
    def __init__(self, output_dim: int = 256, num_classes: int = 3):
        super().__init__(output_dim, num_classes)


class NuScenesExpertExtractor(nn.Module):
    """expert_extractors.py:108-137."""

    def __init__(self, output_dim: int = 256, num_queries: int = 100, num_classes: int = 10, bbox_dim: int = 7):
        super().__init__()
        self.output_dim, self.num_queries, self.num_classes, self.bbox_dim = output_dim, num_queries, num_classes, bbox_dim
        self.feature_extractor = nn.Sequential(nn.Linear(num_queries * (num_classes + bbox_dim), 512), nn.ReLU(), nn.Dropout(0.1),
                                               nn.Linear(512, output_dim), nn.LayerNorm(output_dim))

    def forward(self, o):
        c = torch.cat([o["class_logits"], o["bbox_preds"]], dim=-1)
        return self.feature_extractor(c.view(c.size(0), -1))


class ExpertOutputManager(nn.Module):
    def __init__(self, extractors):
        super().__init__()
        self.extractors = nn.ModuleList(extractors)

    def extract_features(self, outs):
        return [e(o) for e, o in zip(self.extractors, outs)]


def create_expert_extractors(expert_configs: List[Dict]) -> ExpertOutputManager:
    table = {"detection": (DetectionExpertExtractor, 10), "segmentation": (SegmentationExpertExtractor, 19),
             "drivable": (DrivableExpertExtractor, 3)}
    ex = []
    for c in expert_configs:
        if c["type"] == "nuscenes":
            ex.append(NuScenesExpertExtractor(output_dim=c.get("output_dim", 256), num_queries=c.get("num_queries", 100),
                                              num_classes=c.get("num_classes", 10), bbox_dim=c.get("bbox_dim", 7)))
            continue
        if c["type"] not in table:
            raise ValueError(f"Unknown expert type: {c['type']}")
        cls, ncls = table[c["type"]]
        ex.append(cls(output_dim=c.get("output_dim", 256), num_classes=c.get("num_classes", ncls)))
    return ExpertOutputManager(ex)


# --------------------------------------------------------------------------------------------
# Context: models/context/context_features.py:137-165
# --------------------------------------------------------------------------------------------
class SimpleContextExtractor(nn.Module):
    def __init__(self, context_dim: int = 64):
        super().__init__()
        self.context_dim = context_dim
        self.encoder = nn.Sequential(nn.Linear(4, 32), nn.ReLU(), nn.Dropout(0.1), nn.Linear(32, context_dim),
                                     nn.LayerNorm(context_dim))

    def forward(self, speed, steering, throttle, brake):
        return self.encoder(torch.cat([speed, steering, throttle, brake], dim=-1))


# --------------------------------------------------------------------------------------------
# Gating: models/gating/gating_network.py:6-207
# --------------------------------------------------------------------------------------------
class ContextEncoder(nn.Module):
    def __init__(self, context_dim: int = 64, hidden_dim: int = 128):
        super().__init__()
        self.context_encoder = nn.Sequential(nn.Linear(context_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.1),
                                             nn.Linear(hidden_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.1))

    def forward(self, c):
        return self.context_encoder(c)


class ExpertOutputProcessor(nn.Module):
    def __init__(self, expert_output_dim: int, processed_dim: int = 256):
        super().__init__()
        self.processor = nn.Sequential(nn.Linear(expert_output_dim, processed_dim), nn.ReLU(), nn.Dropout(0.1),
                                       nn.Linear(processed_dim, processed_dim), nn.LayerNorm(processed_dim))

    def forward(self, x):
        return self.processor(x)


class GatingNetwork(nn.Module):
    def __init__(self, num_experts: int, context_dim: int = 64, expert_output_dims: Optional[List[int]] = None,
                 processed_dim: int = 256, hidden_dim: int = 128, temperature: float = 1.0,
                 use_softmax: bool = True, top_k: int = 0, noise_type: str = "gumbel", noise_scale: float = 1.0,
                 apply_topk_at_eval: bool = False):
        super().__init__()
        self.num_experts, self.processed_dim = num_experts, processed_dim
        self.temperature, self.use_softmax = temperature, use_softmax
        self.top_k, self.noise_type = max(0, int(top_k)), noise_type
        self.noise_scale, self.apply_topk_at_eval = float(noise_scale), bool(apply_topk_at_eval)
        dims = expert_output_dims if expert_output_dims is not None else [256] * num_experts
        self.context_encoder = ContextEncoder(context_dim, hidden_dim)
        self.expert_processors = nn.ModuleList([ExpertOutputProcessor(d, processed_dim) for d in dims])
        self.gate_network = nn.Sequential(nn.Linear(hidden_dim + processed_dim * num_experts, hidden_dim), nn.ReLU(),
                                          nn.Dropout(0.1), nn.Linear(hidden_dim, num_experts))
        self.output_projection = nn.Linear(processed_dim, processed_dim)

    def _noise(self, shape, device):  # gating_network.py:102-112
        if self.noise_scale <= 0.0:
            return torch.zeros(shape, device=device)
        if self.noise_type.lower() == "gumbel":
            u = torch.rand(shape, device=device).clamp_(1e-6, 1 - 1e-6)
            return -torch.log(-torch.log(u)) * self.noise_scale
        if self.noise_type.lower() == "gaussian":
            return torch.randn(shape, device=device) * self.noise_scale
        return torch.zeros(shape, device=device)

    @staticmethod
    def _topk_mask(logits, k):  # gating_network.py:114-120
        if k <= 0 or k >= logits.size(1):
            return logits
        vals, idx = torch.topk(logits, k, dim=1)
        return torch.full_like(logits, float("-inf")).scatter_(1, idx, vals)

    def _weights(self, logits, apply_topk):
        if apply_topk:
            logits = self._topk_mask(logits + self._noise(logits.shape, logits.device), self.top_k)
        if self.use_softmax:
            return F.softmax(logits / self.temperature, dim=1)
        w = torch.sigmoid(logits)
        return w / (w.sum(dim=1, keepdim=True) + 1e-8)

    def forward(self, expert_outputs, context):  # gating_network.py:122-175
        ctx = self.context_encoder(context)
        proc = [p(x) for x, p in zip(expert_outputs, self.expert_processors)]
        gate_logits = self.gate_network(torch.cat([ctx] + proc, dim=1))
        w = self._weights(gate_logits, self.top_k > 0 and (self.training or self.apply_topk_at_eval))
        combined = torch.zeros(context.size(0), self.processed_dim, device=context.device)
        for i, p in enumerate(proc):  # accumulated in expert order into fp32 zeros, as the reference does
            combined = combined + w[:, i:i + 1] * p
        return {"combined_output": self.output_projection(combined), "expert_weights": w,
                "processed_expert_outputs": proc, "gate_logits": gate_logits}

    def get_gating_logits(self, context):  # gating_network.py:201-207
        ctx = self.context_encoder(context)
        z = torch.zeros(context.size(0), self.processed_dim * self.num_experts, device=context.device)
        return self.gate_network(torch.cat([ctx, z], dim=1))

    def get_expert_weights(self, context):  # gating_network.py:177-199
        return self._weights(self.get_gating_logits(context), self.top_k > 0 and self.apply_topk_at_eval)


# --------------------------------------------------------------------------------------------
# Policy: models/policy/trajectory_head.py:5-63
# --------------------------------------------------------------------------------------------
class EasyBackbone(nn.Module):
    def __init__(self, in_channels: int = 3, out_dim: int = 512):
        super().__init__()
        layers, cin = [], in_channels
        for cout, k in ((32, 5), (64, 3), (128, 3), (256, 3)):
            layers += [nn.Conv2d(cin, cout, k, 2, k // 2), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
            cin = cout
        self.net = nn.Sequential(*layers)
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(256, out_dim)

    def forward(self, x):
        return self.fc(self.pool(self.net(x)).flatten(1))


class TrajectoryPolicy(nn.Module):
    def __init__(self, horizon: int = 8, context_dim: int = 0, backbone_dim: int = 512):
        super().__init__()
        self.horizon = horizon
        self.backbone = EasyBackbone(3, backbone_dim)
        d = backbone_dim + max(context_dim, 0)

        def head(nout):
            return nn.Sequential(nn.Linear(d, 512), nn.ReLU(inplace=True), nn.Linear(512, 512), nn.ReLU(inplace=True),
                                 nn.Linear(512, nout))

        self.head_wp, self.head_spd = head(horizon * 2), head(horizon)

    def forward(self, image, context=None):
        f = self.backbone(image)
        x = f if context is None else torch.cat([f, context], dim=1)
        return {"waypoints": self.head_wp(x).view(-1, self.horizon, 2), "speed": self.head_spd(x).view(-1, self.horizon)}


# --------------------------------------------------------------------------------------------
# AutoMoE composition: models/automoe.py:13-298 (nuScenes expert out of scope)
# --------------------------------------------------------------------------------------------
_EXPERTS = {"detection": (BDDDetectionExpert, 10), "segmentation": (BDDSegmentationExpert, 19),
            "drivable": (BDDDrivableExpert, 3)}


def _last_step(t: torch.Tensor) -> torch.Tensor:  # automoe.py:108-135
    if t.dim() == 2 and t.size(1) > 1:
        return t[:, -1:].contiguous()
    if t.dim() > 2:
        return t.view(t.size(0), -1)[:, -1:].contiguous()
    return t


class AutoMoE(nn.Module):
    def __init__(self, expert_configs, gating_config, context_config, policy_config, device="cpu"):
        super().__init__()
        self.device = device
        self.expert_configs, self.gating_config = expert_configs, gating_config
        self.context_config, self.policy_config = context_config, policy_config
        self.experts = nn.ModuleList()
        for c in expert_configs:
            if c["type"] == "nuscenes":  # automoe.py:63-70
                self.experts.append(NuScenesExpert(num_queries=c.get("num_queries", 100), fusion=c.get("fusion", "concat"),
                                                   use_lidar=c.get("use_lidar", False), use_tnet=c.get("use_tnet", False),
                                                   bbox_dim=c.get("bbox_dim", 7),
                                                   pretrained_backbone=c.get("pretrained_backbone", True)))
                continue
            if c["type"] not in _EXPERTS:
                raise ValueError(f"Unknown expert type: {c['type']}")
            cls, ncls = _EXPERTS[c["type"]]
            self.experts.append(cls(num_classes=c.get("num_classes", ncls),
                                    pretrained_backbone=c.get("pretrained_backbone", True)))
        self.expert_extractors = create_expert_extractors(expert_configs)
        if context_config.get("type", "simple") != "simple":
            raise ValueError(f"Unknown context extractor type: {context_config.get('type')}")
        self.context_extractor = SimpleContextExtractor(context_config.get("context_dim", 64))
        self.gating_network = GatingNetwork(
            num_experts=len(expert_configs), context_dim=context_config.get("context_dim", 64),
            expert_output_dims=[c.get("output_dim", 256) for c in expert_configs],
            processed_dim=gating_config.get("processed_dim", 256), hidden_dim=gating_config.get("hidden_dim", 128),
            temperature=gating_config.get("temperature", 1.0), use_softmax=gating_config.get("use_softmax", True))
        self.policy_head = TrajectoryPolicy(horizon=policy_config.get("num_waypoints", 10),
                                            context_dim=gating_config.get("processed_dim", 256),
                                            backbone_dim=policy_config.get("backbone_dim", 512))
        self.to(device)

    def _context(self, batch):
        speed = batch["speed"]
        speed = speed[:, -1:].contiguous() if speed.dim() == 2 and speed.size(1) > 1 else speed
        if all(k in batch for k in ("speed", "steering", "throttle", "brake")):
            st, th, br = (_last_step(batch[k]) for k in ("steering", "throttle", "brake"))
        else:
            z = torch.zeros(speed.size(0), 1, device=speed.device)
            st, th, br = z, z.clone(), z.clone()
        return self.context_extractor(speed, st, th, br)

    def forward(self, batch):
        ctx = self._context(batch)
        outs = [e({"image": batch["image"], "lidar": batch.get("lidar")}) if isinstance(e, NuScenesExpert) else e(batch["image"])
                for e in self.experts]  # automoe.py:156-187
        feats = self.expert_extractors.extract_features(outs)
        g = self.gating_network(feats, ctx)
        p = self.policy_head(batch["image"], context=g["combined_output"])
        spd = p["speed"]
        return {"waypoints": p["waypoints"], "speed": spd[:, -1:].contiguous() if spd.dim() == 2 else spd,
                "speed_seq": spd, "expert_weights": g["expert_weights"], "expert_outputs": outs,
                "context_features": ctx, "combined_features": g["combined_output"], "gate_logits": g["gate_logits"]}

    def get_expert_weights(self, batch):  # automoe.py:235-238: gate weights from the context alone (zero expert features)
        return self.gating_network.get_expert_weights(self._context(batch))

    def freeze_experts(self):
        for p in self.experts.parameters():
            p.requires_grad = False

    def unfreeze_experts(self):
        for p in self.experts.parameters():
            p.requires_grad = True


def create_automoe_model(config: Dict, device="cpu") -> AutoMoE:
    return AutoMoE(config["experts"], config["gating"], config["context"], config["policy"], device)
