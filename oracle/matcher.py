"""oracle/matcher.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch-CPU fp32 restatement of the reference's Hungarian matcher
(training/hungarian_matcher.py:13-85) and of the two torchvision box ops it uses.

torchvision is absent from the image, so `box_convert` / `generalized_box_iou` are restated
from their published formulas (torchvision.ops.boxes; Rezatofighi et al. 2019 for GIoU).  No
reference test pins them -> "parity unpinned"; they are checked here on hand-computed cases.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_ref", "liblsap_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
        for name, ctype in (("oracle_lsap_f64", ctypes.c_double), ("oracle_lsap_f32", ctypes.c_float)):
            fn = getattr(_LIB, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.POINTER(ctype), ctypes.c_int64, ctypes.c_int64,
                           ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    return _LIB


def lsap_c(cost: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """C restatement (oracle/lsap.c) of scipy.optimize.linear_sum_assignment.  Raises ValueError
    like scipy on NaN / -inf entries or an infeasible matrix."""
    cost = np.ascontiguousarray(cost)
    if cost.ndim != 2:
        raise ValueError("expected a matrix")
    nr, nc = cost.shape
    n = min(nr, nc)
    rows = np.empty(n, dtype=np.int64)
    cols = np.empty(n, dtype=np.int64)
    if n == 0:
        return rows, cols
    if cost.dtype == np.float32:
        fn, ct = _lib().oracle_lsap_f32, ctypes.c_float
    else:
        cost = cost.astype(np.float64, copy=False)
        fn, ct = _lib().oracle_lsap_f64, ctypes.c_double
    rc = fn(cost.ctypes.data_as(ctypes.POINTER(ct)), nr, nc,
            rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), cols.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    if rc == -2:
        raise ValueError("matrix contains invalid numeric entries")
    if rc == -1:
        raise ValueError("cost matrix is infeasible")
    if rc != 0:
        raise MemoryError("oracle lsap failed")
    return rows, cols


# ---- torchvision.ops restated (hungarian_matcher.py:4,49-51; train_bdd100k_ddp.py:12,144) ----
def box_cxcywh_to_xyxy(b: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(b: torch.Tensor) -> torch.Tensor:
    x1, y1, x2, y2 = b.unbind(-1)
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)


def generalized_box_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[N,4] xyxy x [M,4] xyxy -> [N,M].  No eps anywhere (torchvision has none)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b[None, :] - inter
    iou = inter / union
    lt_c = torch.min(a[:, None, :2], b[None, :, :2])
    rb_c = torch.max(a[:, None, 2:], b[None, :, 2:])
    wh_c = (rb_c - lt_c).clamp(min=0)
    area_c = wh_c[..., 0] * wh_c[..., 1]
    return iou - (area_c - union) / area_c


def cost_matrix(pred_logits: torch.Tensor, pred_boxes: torch.Tensor, tgt_labels: torch.Tensor,
                tgt_boxes: torch.Tensor, cost_class: float = 1.0, cost_bbox: float = 5.0,
                cost_giou: float = 2.0) -> torch.Tensor:
    """One image: [Q,C] logits, [Q,D] boxes, [Ni] labels, [Ni,D] boxes -> [Q,Ni] fp32.
    hungarian_matcher.py:36-75 (D == 4 cxcywh, D == 7 BEV, other D without GIoU); summation order bbox, class, giou (:73-75)."""
    prob = pred_logits.softmax(-1)
    c_class = -prob[:, tgt_labels]
    c_bbox = torch.cdist(pred_boxes, tgt_boxes, p=1)
    D = pred_boxes.shape[1]
    if cost_giou > 0 and D == 4:
        c_giou = -generalized_box_iou(box_cxcywh_to_xyxy(pred_boxes), box_cxcywh_to_xyxy(tgt_boxes))
    elif cost_giou > 0 and D == 7:
        # hungarian_matcher.py:52-66: [cx, cy, cz, w, l, h, yaw] -> axis-aligned BEV box (cx -+ w/2, cy -+ l/2)
        def bev(b):
            return torch.stack([b[:, 0] - b[:, 3] / 2, b[:, 1] - b[:, 4] / 2, b[:, 0] + b[:, 3] / 2, b[:, 1] + b[:, 4] / 2], dim=1)
        c_giou = -generalized_box_iou(bev(pred_boxes), bev(tgt_boxes))
    else:
        c_giou = torch.zeros_like(c_bbox)  # hungarian_matcher.py:67-70
    return cost_bbox * c_bbox + cost_class * c_class + cost_giou * c_giou


class HungarianMatcher(torch.nn.Module):
    """hungarian_matcher.py:13-85 (2D boxes).  `solver` is 'c' (oracle/lsap.c) or 'scipy'."""

    def __init__(self, cost_class=1.0, cost_bbox=5.0, cost_giou=2.0, solver: str = "c"):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0
        self.cost_class, self.cost_bbox, self.cost_giou, self.solver = cost_class, cost_bbox, cost_giou, solver

    @torch.no_grad()
    def forward(self, outputs: Dict[str, torch.Tensor], targets: List[Dict[str, torch.Tensor]]):
        out = []
        for b in range(outputs["pred_logits"].shape[0]):
            C = cost_matrix(outputs["pred_logits"][b], outputs["pred_boxes"][b], targets[b]["labels"],
                            targets[b]["boxes"], self.cost_class, self.cost_bbox, self.cost_giou).cpu().numpy()
            if self.solver == "scipy":
                from scipy.optimize import linear_sum_assignment
                r, c = linear_sum_assignment(C)
            else:
                r, c = lsap_c(C)
            out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(c, dtype=torch.int64)))
        return out
