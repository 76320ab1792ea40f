"""Host wrappers for the device-side Hungarian matcher (csrc/lsap.hip)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import lib as _lib
from .conv import ptr, require_hip, stream


def _L():
    return _lib.get()


USE_SPLIT_SOLVER = True  # tests flip this to compare the split solver (am_lsap_batched_ws) with the general kernels


def lsap_batched(cost: torch.Tensor, n_cols: Optional[torch.Tensor] = None, transposed_storage: bool = False):
    """Solve B assignment problems on the device.
    cost: fp32 [B, nr, nc_max] (or, with transposed_storage, [B, nc_max, nr] holding cost[b, j, i]).
    n_cols: int32 [B] valid column counts (default nc_max).
    Returns (row_idx [B,k] int64, col_idx [B,k] int64, count [B] int32, status [B] int32), all on the device."""
    require_hip(cost, "cost matrix")
    cost = cost.contiguous()
    if cost.dtype != torch.float32:
        cost = cost.float()
    if transposed_storage:
        B, nc_max, nr = cost.shape
        rs, cs = 1, nr
    else:
        B, nr, nc_max = cost.shape
        rs, cs = nc_max, 1
    bs = nr * nc_max
    k = max(1, min(nr, nc_max))
    dev = cost.device
    rows = torch.full((B, k), -1, dtype=torch.int64, device=dev)
    cols = torch.full((B, k), -1, dtype=torch.int64, device=dev)
    count = torch.zeros(B, dtype=torch.int32, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    if n_cols is not None:
        n_cols = n_cols.to(device=dev, dtype=torch.int32).contiguous()
    # caller-owned scratch for the split solver (sorted candidate lists + per-image flags): a fresh tensor per call -- a few KB,
    # and inside a captured step it must live in the graph's pool
    import ctypes
    need = ctypes.c_longlong(0)
    _L().am_lsap_batched_workspace_bytes(B, nr, nc_max, ctypes.byref(need))
    if need.value > 0 and USE_SPLIT_SOLVER:
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        _L().am_lsap_batched_ws(ptr(cost) if cost.numel() else None, B, nr, ptr(n_cols), nc_max, bs, rs, cs, ptr(rows), ptr(cols), k,
                                ptr(count), ptr(status), ptr(ws), need.value, stream())
    else:
        _L().am_lsap_batched(ptr(cost) if cost.numel() else None, B, nr, ptr(n_cols), nc_max, bs, rs, cs, ptr(rows), ptr(cols), k,
                             ptr(count), ptr(status), stream())
    return rows, cols, count, status


def match_cost(pred_logits: torch.Tensor, pred_boxes: torch.Tensor, tgt_labels: torch.Tensor, tgt_boxes: torch.Tensor,
               n_tgt: torch.Tensor, w_class: float, w_bbox: float, w_giou: float) -> torch.Tensor:
    """pred_logits [B,Q,C], pred_boxes [B,Q,D], tgt_labels [B,Nmax] int64, tgt_boxes [B,Nmax,D], n_tgt [B] int32 ->
    cost [B,Nmax,Q] fp32 (transposed storage: cost[b, j, q]).  D = 4: cxcywh + GIoU; D = 7: BEV GIoU; else no GIoU term."""
    require_hip(pred_logits, "pred_logits")
    pl = pred_logits.detach().float().contiguous()
    pb = pred_boxes.detach().float().contiguous()
    B, Q, C = pl.shape
    D = pb.shape[2]
    Nmax = tgt_labels.shape[1]
    cost = torch.zeros((B, max(Nmax, 1), Q), dtype=torch.float32, device=pl.device)
    if Nmax > 0:
        assert tgt_boxes.shape[2] == D, "prediction and target boxes must have the same number of box parameters"
        _L().am_match_cost_d(ptr(pl), ptr(pb), D, ptr(tgt_labels.contiguous()), ptr(tgt_boxes.float().contiguous()),
                             ptr(n_tgt.contiguous()), B, Q, C, Nmax, float(w_class), float(w_bbox), float(w_giou), ptr(cost), stream())
    return cost
