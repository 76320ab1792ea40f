"""Expert output extractors -- drop-in for models/experts/expert_extractors.py:5-200."""
from typing import Dict, List

import torch
import torch.nn as nn

from .._nn import Dropout, Flatten, GlobalAvgPoolNCHW, LayerNorm, Linear, MLPSequential, ReLU


class ExpertOutputExtractor(nn.Module):
    def __init__(self, output_dim: int = 256):
        super().__init__()
        self.output_dim = output_dim

    def forward(self, expert_output):
        raise NotImplementedError


def _mlp(cin: int, output_dim: int) -> MLPSequential:
    # indices as the reference: 0 pool, 1 flatten, 2 linear, 3 relu, 4 dropout, 5 linear, 6 layernorm
    return MLPSequential(GlobalAvgPoolNCHW(), Flatten(), Linear(cin, 512), ReLU(), Dropout(0.1), Linear(512, output_dim),
                         LayerNorm(output_dim))


class DetectionExpertExtractor(ExpertOutputExtractor):
    def __init__(self, output_dim: int = 256, num_classes: int = 10):
        super().__init__(output_dim)
        self.num_classes = num_classes
        self.feature_extractor = _mlp(num_classes + 4, output_dim)

    def pre_mlp(self, expert_output: Dict[str, torch.Tensor]) -> torch.Tensor:
        """[B, C] input of the extractor MLP (everything in front of its first Linear)."""
        combined = torch.cat([expert_output["class_logits"], expert_output["bbox_deltas"]], dim=1)
        return self.feature_extractor[1](self.feature_extractor[0](combined))

    def forward(self, expert_output: Dict[str, torch.Tensor]) -> torch.Tensor:
        combined = torch.cat([expert_output["class_logits"], expert_output["bbox_deltas"]], dim=1)
        return self.feature_extractor(combined)


class SegmentationExpertExtractor(ExpertOutputExtractor):
    def __init__(self, output_dim: int = 256, num_classes: int = 19):
        super().__init__(output_dim)
        self.num_classes = num_classes
        self.feature_extractor = _mlp(num_classes, output_dim)

    def pre_mlp(self, expert_output: torch.Tensor) -> torch.Tensor:
        return self.feature_extractor[1](self.feature_extractor[0](expert_output))

    def forward(self, expert_output: torch.Tensor) -> torch.Tensor:
        return self.feature_extractor(expert_output)


class DrivableExpertExtractor(SegmentationExpertExtractor):
    def __init__(self, output_dim: int = 256, num_classes: int = 3):
        super().__init__(output_dim, num_classes)


class NuScenesExpertExtractor(ExpertOutputExtractor):
    """expert_extractors.py:108-137: cat(class_logits, bbox_preds) over the query axis -> flatten -> MLP -> LayerNorm."""

    def __init__(self, output_dim: int = 256, num_queries: int = 100, num_classes: int = 10, bbox_dim: int = 7):
        super().__init__(output_dim)
        self.num_queries, self.num_classes, self.bbox_dim = num_queries, num_classes, bbox_dim
        self.feature_extractor = MLPSequential(Linear(num_queries * (num_classes + bbox_dim), 512), ReLU(), Dropout(0.1),
                                               Linear(512, output_dim), LayerNorm(output_dim))

    def pre_mlp(self, expert_output: Dict[str, torch.Tensor]) -> torch.Tensor:
        combined = torch.cat([expert_output["class_logits"], expert_output["bbox_preds"]], dim=-1)
        return combined.view(combined.size(0), -1)

    def forward(self, expert_output: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self.feature_extractor(self.pre_mlp(expert_output))


class ExpertOutputManager(nn.Module):
    def __init__(self, extractors: List[ExpertOutputExtractor]):
        super().__init__()
        self.extractors = nn.ModuleList(extractors)

    def extract_features(self, expert_outputs) -> List[torch.Tensor]:
        return [ex(out) for ex, out in zip(self.extractors, expert_outputs)]


def create_expert_extractors(expert_configs: List[Dict]) -> ExpertOutputManager:
    table = {"detection": (DetectionExpertExtractor, 10), "segmentation": (SegmentationExpertExtractor, 19),
             "drivable": (DrivableExpertExtractor, 3)}
    extractors = []
    for config in expert_configs:
        t = config["type"]
        if t == "nuscenes":
            extractors.append(NuScenesExpertExtractor(output_dim=config.get("output_dim", 256), num_queries=config.get("num_queries", 100),
                                                      num_classes=config.get("num_classes", 10), bbox_dim=config.get("bbox_dim", 7)))
            continue
        if t not in table:
            raise ValueError(f"Unknown expert type: {t}")
        cls, ncls = table[t]
        extractors.append(cls(output_dim=config.get("output_dim", 256), num_classes=config.get("num_classes", ncls)))
    return ExpertOutputManager(extractors)
