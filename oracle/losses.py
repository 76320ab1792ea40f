"""oracle/losses.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch-CPU fp32 restatement of the loss assembly on the hot path:
  detection_set_loss   training/train_bdd100k_ddp.py:117-186
  segmentation_loss    training/train_bdd100k_ddp.py:58,188-194
  gating_losses        training/train_gating_network.py:21-74
  train_step           the step glue of train_bdd100k_ddp.py:89-100 / train_gating_network.py:92-105
  policy_losses, carla_detection_loss, carla_sanitize_mask   train_carla_policy.py:22-30, train_carla_bdd_experts_ddp.py:71-140
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from .matcher import HungarianMatcher, box_xyxy_to_cxcywh


def detection_set_loss(model_out: Dict[str, torch.Tensor], gt_boxes: torch.Tensor, gt_labels: torch.Tensor,
                       num_classes: int, matcher: HungarianMatcher, bbox_loss_weight: float = 2.0):
    """gt_boxes [B,Nmax,4] xyxy pixels padded with -1, gt_labels [B,Nmax] padded with -1.
    Returns (total, class_loss, bbox_loss, indices)."""
    logits, boxes = model_out["class_logits"], model_out["bbox_deltas"]
    B, C, H, W = logits.shape
    Q = H * W
    logits = logits.permute(0, 2, 3, 1).reshape(B, Q, C)
    boxes = boxes.permute(0, 2, 3, 1).reshape(B, Q, 4)
    targets = []
    for b in range(B):
        keep = gt_labels[b] != -1
        bx = gt_boxes[b][keep]
        targets.append({"boxes": box_xyxy_to_cxcywh(bx) if bx.numel() > 0 else bx, "labels": gt_labels[b][keep]})
    indices = matcher({"pred_logits": logits, "pred_boxes": boxes}, targets)
    tgt_cls = torch.full((B * Q,), num_classes, dtype=torch.int64)
    tgt_box = torch.zeros((B * Q, 4), dtype=torch.float32)
    for b, (pi, ti) in enumerate(indices):
        tgt_cls[b * Q + pi] = targets[b]["labels"][ti]
        tgt_box[b * Q + pi] = targets[b]["boxes"][ti]
    cls_loss = F.cross_entropy(logits.reshape(B * Q, C), tgt_cls, ignore_index=num_classes)
    matched = tgt_cls != num_classes
    if matched.any():
        box_loss = F.smooth_l1_loss(boxes.reshape(B * Q, 4)[matched], tgt_box[matched], reduction="mean")
    else:
        box_loss = torch.tensor(0.0)
    return cls_loss + bbox_loss_weight * box_loss, cls_loss, box_loss, indices


def segmentation_loss(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    return F.cross_entropy(logits, mask, ignore_index=255)


def gating_losses(pred: Dict[str, torch.Tensor], target_wp: torch.Tensor, target_spd: torch.Tensor,
                  config: Dict) -> Dict[str, torch.Tensor]:
    wp = pred["waypoints"]
    ade = F.l1_loss(wp, target_wp)
    fde = F.l1_loss(wp[:, -1, :], target_wp[:, -1, :])
    ps = pred.get("speed_seq", pred.get("speed"))
    if ps is not None and ps.dim() == 2 and target_spd.dim() == 2 and ps.size(1) == target_spd.size(1):
        spd = F.l1_loss(ps, target_spd)
    else:
        pl = pred.get("speed")
        if pl is not None and pl.dim() == 2 and pl.size(1) == 1:
            spd = F.l1_loss(pl, target_spd[:, -1:].contiguous())
        else:
            spd = torch.zeros(())
    d = wp[:, 1:, :] - wp[:, :-1, :]
    smooth = F.l1_loss(d[:, 1:, :], d[:, :-1, :])
    w = pred["expert_weights"]
    if config.get("use_load_balancing", True):
        usage = w.mean(dim=0)
        lb = F.mse_loss(usage, torch.ones_like(usage) / usage.size(0))
    else:
        lb = torch.tensor(0.0)
    if config.get("use_entropy_loss", True):
        ent = (w * torch.log(w + 1e-8)).sum(dim=1).mean()  # = -entropy
    else:
        ent = torch.tensor(0.0)
    total = (config.get("ade_weight", 1.0) * ade + config.get("fde_weight", 2.0) * fde
             + config.get("speed_weight", 0.2) * spd + config.get("smoothness_weight", 0.1) * smooth
             + config.get("load_balancing_weight", 0.01) * lb + config.get("entropy_weight", 0.001) * ent)
    return {"total_loss": total, "ade": ade, "fde": fde, "speed": spd, "smoothness": smooth,
            "load_balancing": lb, "entropy": ent}


def clip_and_step(params: List[torch.Tensor], optimizer: torch.optim.Optimizer, max_norm: float = 1.0):
    """clip_grad_norm_(max_norm) + optimizer.step(); returns the pre-clip total norm."""
    norm = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)
    optimizer.step()
    return norm


# ---- the CARLA trainers' loss glue ----
def policy_losses(pred: Dict[str, torch.Tensor], target_wp: torch.Tensor, target_spd: torch.Tensor) -> Dict[str, torch.Tensor]:
    """training/train_carla_policy.py:22-30."""
    ade = F.l1_loss(pred["waypoints"], target_wp)
    fde = F.l1_loss(pred["waypoints"][:, -1, :], target_wp[:, -1, :])
    l_spd = F.l1_loss(pred["speed"], target_spd)
    d = pred["waypoints"][:, 1:, :] - pred["waypoints"][:, :-1, :]
    l_smooth = F.l1_loss(d[:, 1:, :], d[:, :-1, :])
    return {"loss": ade + 2.0 * fde + 0.2 * l_spd + 0.1 * l_smooth, "ade": ade, "fde": fde, "speed": l_spd, "smooth": l_smooth}


def carla_detection_loss(model_out: Dict[str, torch.Tensor], gt_boxes: torch.Tensor, gt_labels: torch.Tensor, num_classes: int,
                         matcher: HungarianMatcher, bbox_loss_weight: float = 1.0):
    """training/train_carla_bdd_experts_ddp.py:71-127: class loss = mean CE over the matched queries (0.0 without matches),
    SmoothL1(mean) over the matched boxes, total = cls + bbox_loss_weight * box."""
    logits, boxes = model_out["class_logits"], model_out["bbox_deltas"]
    B, C, H, W = logits.shape
    Q = H * W
    logits = logits.permute(0, 2, 3, 1).reshape(B, Q, C)
    boxes = boxes.permute(0, 2, 3, 1).reshape(B, Q, 4)
    targets = []
    for b in range(B):
        keep = gt_labels[b] != -1
        bx = gt_boxes[b][keep]
        targets.append({"boxes": box_xyxy_to_cxcywh(bx) if bx.numel() > 0 else bx, "labels": gt_labels[b][keep]})
    indices = matcher({"pred_logits": logits, "pred_boxes": boxes}, targets)
    tgt_cls = torch.full((B * Q,), num_classes, dtype=torch.int64)
    tgt_box = torch.zeros((B * Q, 4), dtype=torch.float32)
    for b, (pi, ti) in enumerate(indices):
        if pi.numel() > 0:
            tgt_cls[b * Q + pi] = targets[b]["labels"][ti]
            tgt_box[b * Q + pi] = targets[b]["boxes"][ti]
    valid = tgt_cls != num_classes
    if valid.any():
        cls_loss = F.cross_entropy(logits.reshape(B * Q, C)[valid], tgt_cls[valid], reduction="mean")
        box_loss = F.smooth_l1_loss(boxes.reshape(B * Q, 4)[valid], tgt_box[valid], reduction="mean")
    else:
        cls_loss, box_loss = torch.tensor(0.0), torch.tensor(0.0)
    return cls_loss + bbox_loss_weight * box_loss, cls_loss, box_loss, indices


def carla_sanitize_mask(mask: torch.Tensor, num_classes: int) -> torch.Tensor:
    """training/train_carla_bdd_experts_ddp.py:132-138."""
    if mask.dim() == 4:
        mask = mask[..., 0]
    invalid = (mask < 0) | (mask >= num_classes)
    if invalid.any():
        mask = mask.clone()
        mask[invalid] = 255
    return mask


def nuscenes_set_loss(model_out: Dict[str, torch.Tensor], gt_boxes: torch.Tensor, gt_labels: torch.Tensor, matcher: HungarianMatcher,
                      bbox_loss_weight: float = 5.0):
    """training/train_nuscenes_expert_ddp.py:73-112: CE(ignore_index=-1) over all queries (unmatched = -1); SmoothL1('none')
    of ALL query boxes vs a target that is zero for unmatched queries, mean over every element."""
    logits, boxes = model_out["class_logits"], model_out["bbox_preds"]
    B, Q, C = logits.shape
    targets = []
    for i in range(B):
        keep = gt_labels[i] != -1
        targets.append({"boxes": gt_boxes[i][keep], "labels": gt_labels[i][keep]})
    indices = matcher({"pred_logits": logits, "pred_boxes": boxes}, targets)
    tgt_classes = torch.full((B, Q), -1, dtype=torch.int64)
    tgt_boxes = torch.zeros_like(boxes)
    for i, (pi, ti) in enumerate(indices):
        tgt_classes[i, pi] = targets[i]["labels"][ti]
        tgt_boxes[i, pi] = targets[i]["boxes"][ti]
    loss_cls = F.cross_entropy(logits.view(-1, C), tgt_classes.view(-1), ignore_index=-1)
    loss_bbox = F.smooth_l1_loss(boxes, tgt_boxes, reduction="none").mean()
    return loss_cls + bbox_loss_weight * loss_bbox, loss_cls, loss_bbox, indices


def detection_val_metrics(pred_boxes: torch.Tensor, targets_cxcywh: List[Dict[str, torch.Tensor]], indices) -> Dict[str, float]:
    """training/train_bdd100k_ddp.py:267-291 as written there: per-image Python loops, `.item()` per image.
    pred_boxes [B,Q,4]; targets_cxcywh: per image {'boxes' [Ni,4]}; indices: the matcher's per-image (pred_idx, tgt_idx)."""
    from .matcher import box_cxcywh_to_xyxy

    def box_iou(a, b):
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[None, :, :2])
        rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        return inter / (area_a[:, None] + area_b[None, :] - inter)

    iou_scores, recall_vals = [], []
    for b, (p_idx, t_idx) in enumerate(indices):
        if p_idx.numel() > 0:
            pr = pred_boxes[b][p_idx]
            gt = targets_cxcywh[b]["boxes"][t_idx]
            iou_scores.append(box_iou(box_cxcywh_to_xyxy(pr), box_cxcywh_to_xyxy(gt)).diagonal().mean().item())
        if t_idx.numel() > 0:
            mat = box_iou(box_cxcywh_to_xyxy(pred_boxes[b]), box_cxcywh_to_xyxy(targets_cxcywh[b]["boxes"]))
            recall_vals.append((mat.max(dim=0)[0] >= 0.5).float().mean().item())
    return {"avg_iou": float(sum(iou_scores) / len(iou_scores)) if iou_scores else 0.0,
            "recall_0.5": float(sum(recall_vals) / len(recall_vals)) if recall_vals else 0.0}


def segmentation_val_metrics(outputs: torch.Tensor, masks: torch.Tensor) -> Dict[str, float]:
    """training/train_bdd100k_ddp.py:299-325 as written there (loop over classes, `.item()` per class)."""
    preds = outputs.argmax(dim=1)
    ignore_mask = masks == 255
    valid = ~ignore_mask
    correct = (preds == masks) & valid
    pixel_acc = correct.sum().float() / valid.sum().clamp(min=1).float()
    ious = []
    for cls in range(outputs.shape[1]):
        gt_cls = masks == cls
        if gt_cls.sum() == 0:
            continue
        pred_cls = preds == cls
        inter = (pred_cls & gt_cls).sum().float()
        union = ((pred_cls | gt_cls) & ~ignore_mask).sum().float()
        ious.append((inter / union).item())
    return {"pixel_acc": pixel_acc.item(), "mean_iou": sum(ious) / len(ious) if ious else 0.0}
