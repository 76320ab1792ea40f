"""CARLA fine-tuning of the BDD experts -- drop-in for training/train_carla_bdd_experts_ddp.py (Trainer :40-185, main :243-270)
on the MI355X HIP path.

Same kernels and step glue as training/train_bdd100k_ddp.py; what differs is the loss glue, kept as in the reference:
  detection (:71-127)   class loss = mean cross-entropy over the MATCHED queries, 0.0 when nothing matched; SmoothL1(mean)
                        on the matched boxes; total = cls + bbox_loss_weight * box with bbox_loss_weight default 1.0
  segmentation (:129-140) labels outside [0, num_classes) become 255 (ignored); a trailing channel axis of the mask is dropped
  step (:142-162)       AdamW + per-step CosineAnnealingLR + clip 1.0; validation reuses the training loss (:171-185);
                        one checkpoint `best.pth` {model_state_dict, best_val_loss, config} at the end of the run (:224-240)
`--synthetic` replaces the CARLA loaders (dataset absent offline) with same-layout batches.
"""
import argparse
import json
import os
from pathlib import Path

import torch
import torch.distributed as dist

from ..hip import ops as hops
from ..models.experts import BDDDetectionExpert, BDDDrivableExpert, BDDSegmentationExpert
from . import synthetic
from .ddp import DataParallel
from .train_bdd100k_ddp import BDDTrainer, detection_set_loss


def sanitize_mask(mask: torch.Tensor, num_classes: int) -> torch.Tensor:
    """train_carla_bdd_experts_ddp.py:132-138, without the data-dependent branch (no host sync)."""
    if mask.dim() == 4:
        mask = mask[..., 0]
    return torch.where((mask < 0) | (mask >= num_classes), torch.full_like(mask, 255), mask)


class Trainer(BDDTrainer):
    def __init__(self, task, model, train_loader, val_loader, device, config, rank: int = 0):
        super().__init__("detection" if task == "detection" else task, model, train_loader, val_loader, device, config)
        self.rank = rank
        if getattr(self.core, "num_classes", None) is None:
            raise AttributeError(("Detection" if task == "detection" else "Segmentation") + " model must expose num_classes")

    def _train_detection_batch(self, batch):
        out = self.model(batch["image"].to(self.device))
        total, _, _, _ = detection_set_loss(out, batch["bboxes"].to(self.device), batch["labels"].to(self.device), self.core.num_classes,
                                            self.matcher, self.config.get("bbox_loss_weight", 1.0), zero_when_unmatched=True)
        return total

    def _train_segmentation_batch(self, batch):
        mask = sanitize_mask(batch["mask"].to(self.device), self.core.num_classes)
        return self._segmentation_loss(batch["image"].to(self.device), mask.contiguous())

    @torch.no_grad()
    def validate(self, epoch=None):
        """train_carla_bdd_experts_ddp.py:171-184: the mean of the TRAINING objective over the validation loader (no metrics, no
        epoch argument); accumulated on the device, read once."""
        self.model.eval()
        total = torch.zeros((), device=self.device)
        n = 0
        for batch in self.val_loader:
            total += self._train_detection_batch(batch) if self.task == "detection" else self._train_segmentation_batch(batch)
            n += 1
        t = torch.stack([total, torch.tensor(float(n), device=self.device)])
        if dist.is_initialized():
            dist.all_reduce(t)
        return float(t[0].item()) / max(1.0, float(t[1].item()))

    def save_best(self, epoch, val_loss):
        if dist.is_initialized() and dist.get_rank() != 0:
            return
        ckpt_dir = Path(f"models/checkpoints/carla_{self.task}_expert_ddp/{self.config['run_name']}")
        ckpt_dir.mkdir(parents=True, exist_ok=True)
        torch.save({"model_state_dict": self.core.state_dict(), "best_val_loss": val_loss, "config": self.config}, ckpt_dir / "best.pth")


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="DDP: Fine-tune BDD experts on CARLA (MI355X HIP path)")
    p.add_argument("--task", type=str, required=True, choices=["detection", "segmentation", "drivable"])
    p.add_argument("--data_root", type=str, default="datasets/carla/preprocessed")
    p.add_argument("--epochs", type=int, default=20)
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--learning_rate", type=float, default=2e-4)
    p.add_argument("--weight_decay", type=float, default=1e-5)
    p.add_argument("--bbox_loss_weight", type=float, default=1.0)
    p.add_argument("--cost_class", type=float, default=1.0)
    p.add_argument("--cost_bbox", type=float, default=5.0)
    p.add_argument("--cost_giou", type=float, default=2.0)
    p.add_argument("--run_name", type=str, default="carla_ft_ddp")
    p.add_argument("--synthetic", action="store_true", help="synthetic CARLA-shaped batches (no dataset offline)")
    p.add_argument("--synthetic_steps", type=int, default=20)
    p.add_argument("--pretrained_backbone", action="store_true", help="needs AUTOMOE_RESNET18_WEIGHTS (the reference fetches)")
    p.add_argument("--precision", choices=["fp16", "fp32"], default="fp16")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    from .. import runtime
    runtime.set_compute_dtype(torch.float16 if args.precision == "fp16" else torch.float32)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dist.init_process_group(backend="nccl", init_method="env://")  # RCCL
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cuda")
    if not args.synthetic:
        raise SystemExit("the CARLA dataset is not available offline; run with --synthetic")
    rank = int(os.environ.get("RANK", "0"))
    cls = {"detection": BDDDetectionExpert, "segmentation": BDDSegmentationExpert, "drivable": BDDDrivableExpert}[args.task]
    model = cls(pretrained_backbone=args.pretrained_backbone).to(device)
    if args.task == "detection":
        batch = synthetic.bdd_detection_batch(args.batch_size, device=device, seed=rank)
    else:
        batch = synthetic.bdd_drivable_batch(args.batch_size, num_classes=model.num_classes, device=device, seed=rank)
    loader = synthetic.SyntheticLoader(batch, args.synthetic_steps)
    wrapped = DataParallel(model) if dist.is_initialized() else model
    trainer = Trainer(args.task, wrapped, loader, loader, device, vars(args), rank)
    best = float("inf")
    for epoch in range(args.epochs):
        trainer.train_epoch(epoch)
        best = min(best, trainer.validate())
    trainer.save_best(args.epochs, best)
    if rank == 0:
        cfg_dir = Path(f"models/configs/carla_{args.task}_expert_ddp")
        cfg_dir.mkdir(parents=True, exist_ok=True)
        with open(cfg_dir / f"{args.run_name}_config.json", "w") as f:
            json.dump(vars(args), f, indent=2)
        print(f"completed {args.epochs} epochs | best_val_loss={best:.4f}")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
