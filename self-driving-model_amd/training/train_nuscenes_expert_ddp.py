"""NuScenes expert trainer -- drop-in for training/train_nuscenes_expert_ddp.py (loss :73-112, validate :132-185, flags :243-257)
on the MI355X HIP path (image-only expert).

Loss glue kept as in the reference: Hungarian matching of the Q queries against the padded ground truth ([B,M,7] boxes,
[B,M] labels, -1 padding) with the D = 7 BEV-GIoU cost; class loss = CrossEntropy(ignore_index=-1) over all queries with
unmatched targets -1 (= mean over the matched ones); box loss = SmoothL1(reduction='none') of ALL query boxes against a
target tensor that is zero for unmatched queries, `.mean()` over every element; total = cls + bbox_loss_weight (5.0) * box.
The matcher and the target scatter run on the device without host synchronisation.
"""
import argparse
import os
from pathlib import Path

import torch
import torch.distributed as dist

from ..hip import ops as hops
from ..models.experts import NuScenesExpert
from . import synthetic
from .ddp import DataParallel
from .hungarian_matcher import HungarianMatcher
from .train_bdd100k_ddp import BDDTrainer


def nuscenes_set_loss(outputs, gt_boxes, gt_labels, matcher: HungarianMatcher, bbox_loss_weight: float = 5.0):
    """train_nuscenes_expert_ddp.py:73-112.  outputs {'class_logits' [B,Q,C], 'bbox_preds' [B,Q,D]}; gt_boxes [B,M,D],
    gt_labels [B,M] padded with -1 (trailing).  Returns (total, class_loss, bbox_loss, match)."""
    logits, boxes = outputs["class_logits"], outputs["bbox_preds"]
    B, Q, C = logits.shape
    D = boxes.shape[2]
    dev = logits.device
    n_tgt = (gt_labels != -1).sum(dim=1).to(torch.int32)
    rows, cols, count, status = matcher.match_padded(logits, boxes, gt_labels, gt_boxes.float(), n_tgt)
    k = rows.shape[1]
    ok = torch.arange(k, device=dev)[None, :] < count[:, None]
    r = rows.clamp(min=0)
    c = cols.clamp(min=0, max=max(gt_labels.shape[1] - 1, 0))
    flat = torch.where(ok, torch.arange(B, device=dev)[:, None] * Q + r, torch.full_like(r, B * Q)).reshape(-1)  # dummy last slot
    tgt_cls = torch.full((B * Q + 1,), -1, dtype=torch.int64, device=dev)
    tgt_box = torch.zeros((B * Q + 1, D), dtype=torch.float32, device=dev)
    tgt_cls.scatter_(0, flat, torch.gather(gt_labels, 1, c).reshape(-1))
    tgt_box.scatter_(0, flat[:, None].expand(-1, D), torch.gather(gt_boxes.float(), 1, c[..., None].expand(-1, -1, D)).reshape(-1, D))
    tgt_cls, tgt_box = tgt_cls[: B * Q].clone(), tgt_box[: B * Q]
    # CrossEntropy(ignore_index=-1) over the B*Q rows: the pixel-CE kernel on a [1, C, B*Q, 1] view (ignore value remapped to C)
    ce_tgt = torch.where(tgt_cls < 0, torch.full_like(tgt_cls, C), tgt_cls).view(1, B * Q, 1)
    class_loss = hops.CrossEntropy2d.apply(logits.reshape(1, B * Q, C).permute(0, 2, 1).unsqueeze(-1).contiguous(), ce_tgt, C)
    d = (boxes.reshape(B * Q, D) - tgt_box).abs()
    bbox_loss = torch.where(d < 1.0, 0.5 * d * d, d - 0.5).mean()
    return class_loss + bbox_loss_weight * bbox_loss, class_loss, bbox_loss, (rows, cols, count, status)


def nuscenes_batch(B: int, H: int = 720, W: int = 1280, max_boxes: int = 16, bbox_dim: int = 7, device="cuda", seed: int = 0):
    """Synthetic batch in the nuScenes loader's layout: image, lidar [B,P,3] (unused by the image-only expert), boxes
    [B,M,bbox_dim] and labels [B,M] padded with -1."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    counts = torch.randint(1, max_boxes + 1, (B,), device=device, generator=g)
    boxes = torch.randn(B, max_boxes, bbox_dim, device=device, generator=g) * 10
    if bbox_dim >= 5:
        boxes[..., 3:5] = boxes[..., 3:5].abs() + 0.5
    labels = torch.randint(0, 10, (B, max_boxes), device=device, generator=g, dtype=torch.int64)
    pad = torch.arange(max_boxes, device=device)[None, :] >= counts[:, None]
    boxes[pad] = -1.0
    labels[pad] = -1
    return {"image": torch.randn(B, 3, H, W, device=device, generator=g), "lidar": torch.zeros(B, 8, 3, device=device),
            "intrinsics": torch.eye(3, device=device).expand(B, 3, 3).contiguous(), "boxes": boxes, "labels": labels}


class NuScenesTrainer(BDDTrainer):
    def __init__(self, model, train_loader, val_loader, device, config):
        super().__init__("detection", model, train_loader, val_loader, device, config)

    def _train_detection_batch(self, batch):
        out = self.model({"image": batch["image"].to(self.device), "lidar": batch.get("lidar")})
        total, _, _, _ = nuscenes_set_loss(out, batch["boxes"].to(self.device), batch["labels"].to(self.device), self.matcher,
                                           self.config.get("bbox_loss_weight", 5.0))
        return total

    @torch.no_grad()
    def _evaluate_detection_batch(self, batch):
        # validation objective = the training objective on the validation batches (the BDD box metrics assume 4-number boxes)
        return self._train_detection_batch(batch), {}

    def save_best(self, epoch, val_loss):
        if dist.is_initialized() and dist.get_rank() != 0:
            return
        ckpt_dir = Path(f"models/checkpoints/nuscenes_expert/{self.config['run_name']}")
        ckpt_dir.mkdir(parents=True, exist_ok=True)
        torch.save({"epoch": epoch, "model_state_dict": self.core.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
                    "scheduler_state_dict": self.scheduler.state_dict(), "best_val_loss": val_loss, "config": self.config},
                   ckpt_dir / "best_full.pth")
        torch.save(self.core.state_dict(), ckpt_dir / "best_model.pth")


def main(argv=None):
    p = argparse.ArgumentParser(description="Train NuScenes Expert (MI355X HIP path, image branch)")
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--num_workers", type=int, default=4)
    p.add_argument("--run_name", type=str, default="run_001")
    p.add_argument("--cost_class", type=float, default=1.0)
    p.add_argument("--cost_bbox", type=float, default=5.0)
    p.add_argument("--cost_giou", type=float, default=2.0)
    p.add_argument("--bbox_loss_weight", type=float, default=5.0)
    p.add_argument("--num_queries", type=int, default=100)
    p.add_argument("--resume_from", type=str, default="")
    p.add_argument("--resume_mode", type=str, choices=["model", "full"], default="model")
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--synthetic", action="store_true", help="synthetic nuScenes-shaped batches (no dataset offline)")
    p.add_argument("--synthetic_steps", type=int, default=20)
    p.add_argument("--pretrained_backbone", action="store_true", help="needs AUTOMOE_RESNET18_WEIGHTS (the reference fetches)")
    p.add_argument("--precision", choices=["fp16", "fp32"], default="fp16")
    args = p.parse_args(argv)
    from .. import runtime
    runtime.set_compute_dtype(torch.float16 if args.precision == "fp16" else torch.float32)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dist.init_process_group(backend="nccl", init_method="env://")  # RCCL
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device(args.device)
    if not args.synthetic:
        raise SystemExit("nuScenes is not available offline; run with --synthetic")
    rank = int(os.environ.get("RANK", "0"))
    model = NuScenesExpert(num_queries=args.num_queries, pretrained_backbone=args.pretrained_backbone).to(device)
    loader = synthetic.SyntheticLoader(nuscenes_batch(args.batch_size, device=device, seed=rank), args.synthetic_steps)
    wrapped = DataParallel(model) if dist.is_initialized() else model
    trainer = NuScenesTrainer(wrapped, loader, loader, device, vars(args))
    if args.resume_from:
        ck = torch.load(args.resume_from, map_location=device, weights_only=True)
        trainer.core.load_state_dict(ck.get("model_state_dict", ck) if isinstance(ck, dict) else ck, strict=True)
        if args.resume_mode == "full" and isinstance(ck, dict):
            trainer.load_training_state(ck)
    trainer.train()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
