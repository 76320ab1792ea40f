"""Deterministic, platform-independent weights/inputs for fixtures and parity tests.

Weights are drawn from numpy's PCG64 (bit-stable across machines), not torch's RNG, so a fixture
only has to store a seed instead of a state_dict.
"""
from __future__ import annotations

import numpy as np
import torch


def seeded_tensor(shape, seed: int, scale: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(tuple(shape)) * scale).astype(np.float32)).to(dtype)


def seed_module_(module: torch.nn.Module, seed: int) -> torch.nn.Module:
    """Fill every parameter/buffer in place, in state_dict order: weights ~ N(0, 1/sqrt(fan_in)),
    1-D params ~ small perturbations around their conventional value, BN running stats non-trivial."""
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if not t.dtype.is_floating_point:
                t.zero_()
                continue
            n = t.numel()
            if t.dim() >= 2:
                fan_in = n // t.shape[0]
                v = rng.standard_normal(n) / np.sqrt(max(fan_in, 1))
            elif name.endswith("running_var"):
                v = 0.5 + rng.random(n)
            elif name.endswith("running_mean"):
                v = 0.1 * rng.standard_normal(n)
            elif name.endswith("weight"):  # norm scales
                v = 1.0 + 0.1 * rng.standard_normal(n)
            else:  # biases
                v = 0.1 * rng.standard_normal(n)
            t.copy_(torch.from_numpy(v.astype(np.float32)).reshape(t.shape))
    return module
