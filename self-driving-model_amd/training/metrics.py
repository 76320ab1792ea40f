"""Validation metrics of the expert trainer -- drop-in for the metric arithmetic of training/train_bdd100k_ddp.py:197-335
(`_evaluate_detection_batch`: mean IoU of the matched pairs and recall at IoU 0.5; `_evaluate_segmentation_batch`: pixel accuracy
and mean IoU over the classes present, ignore_index 255).

Batched tensor arithmetic on the device: no per-image Python loop, no `.item()` per batch (the reference synchronises the host
for every image); the trainer accumulates the returned 0-d tensors and reads them once per epoch.  Validation only: not on the
train-step hot path, so this is torch device glue, not a kernel."""
from typing import Dict

import torch


def _cxcywh_to_xyxy(b: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], dim=-1)


def _pair_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """IoU of xyxy boxes a [..., 4] and b [..., 4] (broadcast), torchvision.ops.box_iou arithmetic (no eps)."""
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    lt = torch.maximum(a[..., :2], b[..., :2])
    rb = torch.minimum(a[..., 2:], b[..., 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a + area_b - inter)


def detection_metrics(pred_boxes: torch.Tensor, tgt_boxes_cxcywh: torch.Tensor, n_tgt: torch.Tensor, rows: torch.Tensor,
                      cols: torch.Tensor, count: torch.Tensor) -> Dict[str, torch.Tensor]:
    """pred_boxes [B,Q,4] (cxcywh, as the head emits them), tgt_boxes_cxcywh [B,Nmax,4] with n_tgt[b] valid rows, and the matcher's
    padded assignment (rows / cols [B,kmax], count [B]).  train_bdd100k_ddp.py:267-291:
      avg_iou    = mean over images WITH a match of (mean IoU of the image's matched prediction / target pairs);
      recall_0.5 = mean over images WITH a match of (fraction of the image's targets whose best IoU over ALL Q predictions >= 0.5).
    Images without a match (no targets) are left out of both means; no image at all -> 0."""
    B, Q, _ = pred_boxes.shape
    dev = pred_boxes.device
    k = rows.shape[1]
    ok = torch.arange(k, device=dev)[None, :] < count[:, None]
    r = rows.clamp(min=0)
    c = cols.clamp(min=0, max=max(tgt_boxes_cxcywh.shape[1] - 1, 0))
    pr = torch.gather(pred_boxes, 1, r[..., None].expand(-1, -1, 4))
    gt = torch.gather(tgt_boxes_cxcywh, 1, c[..., None].expand(-1, -1, 4))
    iou = _pair_iou(_cxcywh_to_xyxy(pr), _cxcywh_to_xyxy(gt))
    iou = torch.where(ok, iou, torch.zeros_like(iou))
    has = count > 0
    per_img = iou.sum(dim=1) / count.clamp(min=1).to(iou.dtype)
    n_img = has.sum().clamp(min=1).to(iou.dtype)
    avg_iou = torch.where(has, per_img, torch.zeros_like(per_img)).sum() / n_img
    # recall: best IoU of every target over all predictions
    valid_t = torch.arange(tgt_boxes_cxcywh.shape[1], device=dev)[None, :] < n_tgt[:, None]
    mat = _pair_iou(_cxcywh_to_xyxy(pred_boxes)[:, :, None, :], _cxcywh_to_xyxy(tgt_boxes_cxcywh)[:, None, :, :])  # [B,Q,Nmax]
    best = mat.max(dim=1)[0]
    hit = ((best >= 0.5) & valid_t).to(iou.dtype).sum(dim=1) / n_tgt.clamp(min=1).to(iou.dtype)
    recall = torch.where(has, hit, torch.zeros_like(hit)).sum() / n_img
    zero = torch.zeros((), device=dev, dtype=iou.dtype)
    return {"avg_iou": torch.where(has.any(), avg_iou, zero), "recall_0.5": torch.where(has.any(), recall, zero)}


def segmentation_metrics(logits: torch.Tensor, masks: torch.Tensor, ignore_index: int = 255) -> Dict[str, torch.Tensor]:
    """train_bdd100k_ddp.py:299-325: pixel accuracy over the non-ignored pixels; mean over the classes PRESENT in the masks of
    |pred == c & gt == c| / |(pred == c | gt == c) & not ignored|."""
    preds = logits.argmax(dim=1)
    valid = masks != ignore_index
    correct = (preds == masks) & valid
    pixel_acc = correct.sum().float() / valid.sum().clamp(min=1).float()
    C = logits.shape[1]
    cls = torch.arange(C, device=logits.device).view(C, 1, 1, 1)
    gt_c = masks[None] == cls
    pr_c = preds[None] == cls
    inter = (pr_c & gt_c).flatten(1).sum(dim=1).float()
    union = ((pr_c | gt_c) & valid[None]).flatten(1).sum(dim=1).float()
    present = gt_c.flatten(1).any(dim=1)
    iou = torch.where(present, inter / union.clamp(min=1), torch.zeros_like(inter))
    mean_iou = iou.sum() / present.sum().clamp(min=1).float()
    return {"pixel_acc": pixel_acc, "mean_iou": torch.where(present.any(), mean_iou, torch.zeros_like(mean_iou))}
