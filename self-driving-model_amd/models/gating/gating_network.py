"""ContextEncoder / ExpertOutputProcessor / GatingNetwork -- drop-in for models/gating/gating_network.py:6-207."""
from typing import Dict, List

import torch
import torch.nn as nn

from ...hip import ops as hops
from .._nn import Dropout, LayerNorm, Linear, MLPSequential, ReLU


class ContextEncoder(nn.Module):
    def __init__(self, context_dim: int = 64, hidden_dim: int = 128):
        super().__init__()
        self.context_dim, self.hidden_dim = context_dim, hidden_dim
        self.context_encoder = MLPSequential(Linear(context_dim, hidden_dim), ReLU(), Dropout(0.1),
                                             Linear(hidden_dim, hidden_dim), ReLU(), Dropout(0.1))

    def forward(self, context: torch.Tensor) -> torch.Tensor:
        return self.context_encoder(context)


class ExpertOutputProcessor(nn.Module):
    def __init__(self, expert_output_dim: int, processed_dim: int = 256):
        super().__init__()
        self.expert_output_dim, self.processed_dim = expert_output_dim, processed_dim
        self.processor = MLPSequential(Linear(expert_output_dim, processed_dim), ReLU(), Dropout(0.1),
                                       Linear(processed_dim, processed_dim), LayerNorm(processed_dim))

    def forward(self, expert_output: torch.Tensor) -> torch.Tensor:
        return self.processor(expert_output)


class GatingNetwork(nn.Module):
    def __init__(self, num_experts: int, context_dim: int = 64, expert_output_dims: List[int] = None, processed_dim: int = 256,
                 hidden_dim: int = 128, temperature: float = 1.0, use_softmax: bool = True, top_k: int = 0,
                 noise_type: str = "gumbel", noise_scale: float = 1.0, apply_topk_at_eval: bool = False):
        super().__init__()
        self.num_experts, self.context_dim = num_experts, context_dim
        self.processed_dim, self.hidden_dim = processed_dim, hidden_dim
        self.temperature, self.use_softmax = temperature, use_softmax
        self.top_k = max(0, int(top_k))
        self.noise_type, self.noise_scale = noise_type, float(noise_scale)
        self.apply_topk_at_eval = bool(apply_topk_at_eval)
        if expert_output_dims is None:
            expert_output_dims = [256] * num_experts
        self.context_encoder = ContextEncoder(context_dim, hidden_dim)
        self.expert_processors = nn.ModuleList([ExpertOutputProcessor(d, processed_dim) for d in expert_output_dims])
        self.gate_network = MLPSequential(Linear(hidden_dim + processed_dim * num_experts, hidden_dim), ReLU(), Dropout(0.1),
                                          Linear(hidden_dim, num_experts))
        self.output_projection = Linear(processed_dim, processed_dim)

    def _sample_noise(self, shape, device):
        if self.noise_scale <= 0.0:
            return None
        if self.noise_type.lower() == "gumbel":
            u = torch.rand(shape, device=device).clamp_(1e-6, 1 - 1e-6)
            return -torch.log(-torch.log(u)) * self.noise_scale
        if self.noise_type.lower() == "gaussian":
            return torch.randn(shape, device=device) * self.noise_scale
        return None

    def _gate(self, gate_logits, processed, apply_topk: bool):
        logits = gate_logits
        k = 0
        if apply_topk:
            noise = self._sample_noise(gate_logits.shape, gate_logits.device)
            if noise is not None:
                logits = gate_logits + noise
            k = self.top_k if self.top_k < self.num_experts else 0
        return hops.GateCombine.apply(logits, self.temperature, self.use_softmax, k, *processed)

    def forward(self, expert_outputs: List[torch.Tensor], context: torch.Tensor) -> Dict[str, torch.Tensor]:
        context_features = self.context_encoder(context)
        processed_outputs = [proc(x) for x, proc in zip(expert_outputs, self.expert_processors)]
        gate_input = torch.cat([context_features] + processed_outputs, dim=1)
        gate_logits = self.gate_network(gate_input)
        apply_topk = (self.top_k > 0) and (self.training or self.apply_topk_at_eval)
        gate_weights, combined = self._gate(gate_logits, processed_outputs, apply_topk)
        return {"combined_output": self.output_projection(combined), "expert_weights": gate_weights,
                "processed_expert_outputs": processed_outputs, "gate_logits": gate_logits}

    def get_gating_logits(self, context: torch.Tensor) -> torch.Tensor:
        context_features = self.context_encoder(context)
        zeros = torch.zeros(context.size(0), self.processed_dim * self.num_experts, device=context.device)
        return self.gate_network(torch.cat([context_features, zeros], dim=1))

    def get_expert_weights(self, context: torch.Tensor) -> torch.Tensor:
        gate_logits = self.get_gating_logits(context)
        dummy = [torch.zeros(context.size(0), self.processed_dim, device=context.device) for _ in range(self.num_experts)]
        weights, _ = self._gate(gate_logits, dummy, (self.top_k > 0) and self.apply_topk_at_eval)
        return weights
