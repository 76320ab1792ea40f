"""Synthetic batch sources with the reference's batch-dict layouts (there are no datasets offline).

  carla_sequence_batch   dataloaders/carla_sequence_loader.py:170-196 (carla_sequence_collate):
                         image [B,3,H,W], speed/steering/throttle/brake [B,T], waypoints [B,T,2]
  bdd_drivable_batch     dataloaders/bdd_drivable_loader.py: image [B,3,H,W], mask [B,H,W] int64 (255 = ignore)
  bdd_detection_batch    dataloaders/bdd_detection_loader.py:11-43 (detection_collate_fn):
                         image, bboxes [B,Nmax,4] xyxy pixels padded with -1, labels [B,Nmax] padded with -1
Definitions follow BASELINE.md section 3 (seed 0, per-rank streams).
"""
from __future__ import annotations

from typing import Dict

import torch


def _gen(device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def carla_sequence_batch(B: int, H: int = 720, W: int = 1280, horizon: int = 10, device="cuda", seed: int = 0) -> Dict[str, torch.Tensor]:
    g = _gen(device, seed)
    r = lambda *s: torch.randn(*s, device=device, generator=g)
    return {"image": r(B, 3, H, W), "speed": r(B, horizon), "steering": r(B, horizon), "throttle": r(B, horizon),
            "brake": r(B, horizon), "waypoints": r(B, horizon, 2)}


def bdd_drivable_batch(B: int, H: int = 720, W: int = 1280, num_classes: int = 3, device="cuda", seed: int = 0, ignore_frac: float = 0.05):
    g = _gen(device, seed)
    img = torch.randn(B, 3, H, W, device=device, generator=g)
    mask = torch.randint(0, num_classes, (B, H, W), device=device, generator=g, dtype=torch.int64)
    mask[torch.rand(B, H, W, device=device, generator=g) < ignore_frac] = 255
    return {"image": img, "mask": mask}


def bdd_detection_batch(B: int, H: int = 720, W: int = 1280, num_classes: int = 10, max_boxes: int = 32, device="cuda", seed: int = 0):
    g = _gen(device, seed)
    img = torch.randn(B, 3, H, W, device=device, generator=g)
    counts = torch.randint(1, max_boxes + 1, (B,), device=device, generator=g)
    nmax = max_boxes
    wh_img = torch.tensor([W, H], device=device, dtype=torch.float32)
    xy = torch.rand(B, nmax, 2, device=device, generator=g) * wh_img * 0.8
    wh = (0.02 + 0.18 * torch.rand(B, nmax, 2, device=device, generator=g)) * wh_img
    boxes = torch.cat([xy, xy + wh], dim=-1)
    labels = torch.randint(0, num_classes, (B, nmax), device=device, generator=g, dtype=torch.int64)
    pad = torch.arange(nmax, device=device)[None, :] >= counts[:, None]
    boxes[pad] = -1.0
    labels[pad] = -1
    return {"image": img, "bboxes": boxes, "labels": labels}


class SyntheticLoader:
    """len()-able iterable of one pre-generated batch (resident in HBM): the timed region never includes H2D."""

    def __init__(self, batch: Dict[str, torch.Tensor], steps: int):
        self.batch, self.steps = batch, steps

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            yield self.batch
