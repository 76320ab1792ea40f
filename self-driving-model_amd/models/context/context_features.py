"""SimpleContextExtractor / create_context_extractor -- drop-in for models/context/context_features.py:137-191.
The 'full' ContextFeatureExtractor is outside the hot path (inconsistent input dims in the reference, unused by
any config: SURVEY.md section 2 row 4)."""
from typing import Dict

import torch
import torch.nn as nn

from .._nn import Dropout, LayerNorm, Linear, MLPSequential, ReLU


class SimpleContextExtractor(nn.Module):
    def __init__(self, context_dim: int = 64):
        super().__init__()
        self.context_dim = context_dim
        self.encoder = MLPSequential(Linear(4, 32), ReLU(), Dropout(0.1), Linear(32, context_dim), LayerNorm(context_dim))

    def forward(self, speed, steering, throttle, brake) -> torch.Tensor:
        vehicle_state = torch.cat([speed, steering, throttle, brake], dim=-1)
        return self.encoder(vehicle_state.float())


def create_context_extractor(config: Dict) -> nn.Module:
    extractor_type = config.get("type", "simple")
    if extractor_type == "simple":
        return SimpleContextExtractor(context_dim=config.get("context_dim", 64))
    if extractor_type == "full":
        raise ValueError("Unknown context extractor type: full (not on the accelerated hot path; see DESIGN.md)")
    raise ValueError(f"Unknown context extractor type: {extractor_type}")
