"""BDD100K expert trainer -- drop-in for training/train_bdd100k_ddp.py (BDDTrainer :26-438, main :440-550) on the
MI355X HIP path.

Kept: CLI flags (:441-457), loss definitions (detection set-loss :117-186 = CE with ignore_index=num_classes over all
queries + SmoothL1(mean) on matched rows, weight 2.0; segmentation/drivable CE(ignore 255) :188-194), AdamW + per-step
CosineAnnealingLR (:39-47,99-100), grad clip 1.0 (:98), checkpoint dict layout (:401-420), resume modes (:536-545).
Changed in HOW: the matcher and the target scatter run on the device with no host sync; clip + AdamW are fused
(training/optim.py); gradients are all-reduced by training/ddp.py; the loss value is accumulated on the device and read
once per epoch; `--synthetic` replaces the BDD100K loaders (dataset absent offline) with same-layout batches.
"""
import argparse
import json
import os
from pathlib import Path

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import runtime
from ..hip import ops as hops
from ..models.experts import BDDDetectionExpert, BDDDrivableExpert, BDDSegmentationExpert
from . import metrics as val_metrics
from . import synthetic
from .ddp import DataParallel, GradBucketReducer, StepStream, capture_error_mode, capture_step, detached
from .hungarian_matcher import HungarianMatcher
from .optim import FusedAdamW


FUSE_SEG_LOSS = True  # tests flip this to compare the fused dense-expert loss with upsample -> CrossEntropy2d


def box_xyxy_to_cxcywh(b: torch.Tensor) -> torch.Tensor:
    x1, y1, x2, y2 = b.unbind(-1)
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)


def detection_set_loss(outputs, gt_boxes, gt_labels, num_classes: int, matcher: HungarianMatcher, bbox_loss_weight: float = 2.0,
                       zero_when_unmatched: bool = False):
    """train_bdd100k_ddp.py:117-186 without host synchronisation.  `zero_when_unmatched` selects the CARLA fine-tuning variant
    (train_carla_bdd_experts_ddp.py:108-118): the class loss is the mean cross-entropy over the matched queries there as
    well, but a batch without any match gives 0.0 instead of the NaN of a cross-entropy whose targets are all ignored.
    outputs: {'class_logits' [B,C,h,w], 'bbox_deltas' [B,4,h,w]}; gt_boxes [B,Nmax,4] xyxy padded with -1; gt_labels [B,Nmax]
    padded with -1 (padding is trailing, as detection_collate_fn produces).  Returns (total, class_loss, bbox_loss, match)."""
    logits, deltas = outputs["class_logits"], outputs["bbox_deltas"]
    B, C, h, w = logits.shape
    Q = h * w
    dev = logits.device
    pred_logits = logits.permute(0, 2, 3, 1).reshape(B, Q, C)
    pred_boxes = deltas.permute(0, 2, 3, 1).reshape(B, Q, 4)
    valid = gt_labels != -1
    n_tgt = valid.sum(dim=1).to(torch.int32)
    tgt_boxes = box_xyxy_to_cxcywh(gt_boxes.float())
    rows, cols, count, status = matcher.match_padded(pred_logits, pred_boxes, gt_labels, tgt_boxes, n_tgt)
    k = rows.shape[1]
    ok = torch.arange(k, device=dev)[None, :] < count[:, None]
    r = rows.clamp(min=0)
    c = cols.clamp(min=0, max=max(gt_labels.shape[1] - 1, 0))
    # masked scatter through a dummy last slot: no boolean indexing, hence no device->host sync
    flat = torch.where(ok, torch.arange(B, device=dev)[:, None] * Q + r, torch.full_like(r, B * Q)).reshape(-1)
    tgt_cls = torch.full((B * Q + 1,), num_classes, dtype=torch.int64, device=dev)
    tgt_box = torch.zeros((B * Q + 1, 4), dtype=torch.float32, device=dev)
    tgt_cls.scatter_(0, flat, torch.gather(gt_labels, 1, c).reshape(-1))
    tgt_box.scatter_(0, flat[:, None].expand(-1, 4), torch.gather(tgt_boxes, 1, c[..., None].expand(-1, -1, 4)).reshape(-1, 4))
    tgt_cls, tgt_box = tgt_cls[: B * Q].clone(), tgt_box[: B * Q]
    tgt_cls = torch.where(tgt_cls < 0, torch.full_like(tgt_cls, num_classes), tgt_cls)
    class_loss = hops.CrossEntropy2d.apply(logits.contiguous(), tgt_cls.view(B, h, w), num_classes)
    matched = (tgt_cls != num_classes)
    if zero_when_unmatched:
        class_loss = torch.where(matched.any(), class_loss, torch.zeros_like(class_loss))
    d = (pred_boxes.reshape(B * Q, 4) - tgt_box).abs()
    sl1 = torch.where(d < 1.0, 0.5 * d * d, d - 0.5) * matched[:, None]
    n_el = (matched.sum() * 4).clamp(min=1)
    bbox_loss = sl1.sum() / n_el  # SmoothL1(mean) over matched rows; 0 when nothing matched (reference: tensor(0.0))
    return class_loss + bbox_loss_weight * bbox_loss, class_loss, bbox_loss, (rows, cols, count, status)


class BDDTrainer:
    def __init__(self, task, model, train_loader, val_loader, device, config):
        self.task, self.model, self.device, self.config = task, model, device, config
        self.core = model.module if hasattr(model, "module") else model
        self.core.to(device)
        self.train_loader, self.val_loader = train_loader, val_loader
        self.optimizer = FusedAdamW(self.core.parameters(), lr=config["learning_rate"], weight_decay=config["weight_decay"], max_norm=1.0)
        self.optimizer.attach_conv_packs(self.core.modules())
        self.reducer = GradBucketReducer(self.optimizer._params, self.optimizer._offsets, self.optimizer.flat_g,
                                         broadcast_from=self.optimizer.flat_p)
        self.optimizer.grad_divisor = float(self.reducer.world)
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=max(1, config["epochs"] * len(train_loader)))
        if task == "detection":
            self.matcher = HungarianMatcher(cost_class=config.get("cost_class", 1.0), cost_bbox=config.get("cost_bbox", 5.0),
                                            cost_giou=config.get("cost_giou", 2.0))
        self.best_val_loss = float("inf")
        # zero_grad + forward + loss (matcher included: cost, assignment and scatter all stay on the device) + backward are
        # ~1000 launches for 9-11 ms of GPU work at the reference's batch sizes: captured into one hipGraph after two eager
        # steps, as in GatingTrainStep.  Batches whose tensor shapes differ from the captured ones (a ragged last batch, a
        # different GT padding) run eagerly.
        self.use_graph = os.environ.get("AUTOMOE_HIPGRAPH", "1") != "0" if config.get("use_graph") is None else bool(config["use_graph"])
        self._graph = self._static = self._static_loss = None
        self._reduce_in_graph = False
        self._eager_steps = 0
        self._step_stream = StepStream(device)  # every step, eager or captured, runs on this stream (training/ddp.py StepStream)

    def load_training_state(self, checkpoint):
        if checkpoint.get("optimizer_state_dict") is not None:
            self.optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
        if checkpoint.get("scheduler_state_dict") is not None:
            self.scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
        self.best_val_loss = float(checkpoint.get("best_val_loss", self.best_val_loss))

    def _train_detection_batch(self, batch):
        images = batch["image"].to(self.device)
        out = self.model(images)
        total, _, _, _ = detection_set_loss(out, batch["bboxes"].to(self.device), batch["labels"].to(self.device),
                                            self.core.num_classes, self.matcher, self.config.get("bbox_loss_weight", 2.0))
        return total

    def _segmentation_loss(self, images, masks):
        """criterion(model(images), masks) of train_bdd100k_ddp.py:89-100.  The dense experts form it from their low-resolution logits
        (pixel_ce_loss: upsample + cross entropy fused, no [B,C,H,W] tensor); FUSE_SEG_LOSS = False keeps the two-op sequence."""
        if FUSE_SEG_LOSS and hasattr(self.core, "pixel_ce_loss"):
            return self.core.pixel_ce_loss(images, masks, 255)
        return hops.CrossEntropy2d.apply(self.model(images), masks, 255)

    def _train_segmentation_batch(self, batch):
        return self._segmentation_loss(batch["image"].to(self.device), batch["mask"].to(self.device))

    def _fwd_bwd(self, batch):
        self.optimizer.zero_grad()
        # conv weight / BatchNorm gradients are added straight into the flat gradient buffer by their kernels (no staging tensor,
        # no accumulate launch per parameter); the kernels' call sites report to the bucketed all-reduce (runtime.grad_ready)
        runtime.set_direct_grads(True)
        try:
            loss = self._train_detection_batch(batch) if self.task == "detection" else self._train_segmentation_batch(batch)
            loss.backward()
        finally:
            runtime.set_direct_grads(False)
        return loss

    def _capture(self, batch):
        self._static = {k: v.to(self.device).clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
        mode = capture_error_mode()  # thread_local: a live process group's watchdog polls events from its own thread (training/ddp.py)

        def capture(in_graph):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._step_stream.stream, capture_error_mode=mode):
                loss = detached(self._fwd_bwd(self._static))
                if in_graph:
                    self.reducer.finish()  # RCCL bucket all-reduces recorded with the step (training/ddp.py)
            return g, loss

        # once the ranks agreed on a mode, every later step -- replayed or (odd batch shape) eager -- exchanges gradients with
        # the same collectives, so ranks whose batches differ in shape still match
        self._graph, self._static_loss, self._reduce_in_graph = capture_step(self.reducer, capture, f"{self.task} train step")
        if self._graph is None:
            self.use_graph = False

    @property
    def input_buffers(self):
        """The captured step's static input tensors (None before the capture): a loader that writes its host-to-device
        copies straight into them (and passes them to train_step) saves a device-to-device copy of the batch per step."""
        return self._static if self._graph is not None else None

    def _fits_graph(self, batch) -> bool:
        return all(isinstance(batch.get(k), torch.Tensor) and batch[k].shape == v.shape and batch[k].dtype == v.dtype
                   for k, v in self._static.items())

    def train_step(self, batch):
        with self._step_stream:
            return self._train_step(batch)

    def _train_step(self, batch):
        dev_ok = torch.device(self.device).type == "cuda"
        if self.use_graph and dev_ok and self._graph is None and self._eager_steps >= 2 and self.core.training:
            self._capture(batch)
        if self._graph is not None and self.core.training and self._fits_graph(batch):
            for k, v in self._static.items():
                if batch[k].data_ptr() != v.data_ptr():
                    v.copy_(batch[k], non_blocking=True)
            self._graph.replay()
            runtime.bump_stats_epoch()  # the replay updated BatchNorm running statistics without running Python
            loss = self._static_loss
            if not self._reduce_in_graph:
                self.reducer.reduce_all()
        elif self._graph is not None:
            self.reducer.reset()
            loss = self._fwd_bwd(batch)
            if self._reduce_in_graph:
                self.reducer.finish()
            else:
                self.reducer.reduce_all()
        else:
            loss = self._fwd_bwd(batch)
            self.reducer.finish()
            self._eager_steps += 1
        self.optimizer.step()  # clip_grad_norm_(1.0) folded in
        self.scheduler.step()
        return detached(loss)

    def train_epoch(self, epoch):
        self.model.train()
        total = torch.zeros((), device=self.device)
        for batch in self.train_loader:
            total += self.train_step(batch).detach()
        return float(total.item()) / max(1, len(self.train_loader))

    @torch.no_grad()
    def _evaluate_detection_batch(self, batch):
        """train_bdd100k_ddp.py:197-295: the validation loss (its box term is divided by the match count a SECOND time there,
        :260-263 -- kept, it decides which checkpoint is "best") and avg IoU / recall@0.5; device tensors, no host sync."""
        out = self.model(batch["image"].to(self.device))
        gt_boxes, gt_labels = batch["bboxes"].to(self.device), batch["labels"].to(self.device)
        _, cls_loss, bbox_loss, (rows, cols, count, _) = detection_set_loss(out, gt_boxes, gt_labels, self.core.num_classes, self.matcher,
                                                                            self.config.get("bbox_loss_weight", 2.0))
        n_matched = count.sum().clamp(min=1).to(bbox_loss.dtype)
        loss = cls_loss + self.config.get("bbox_loss_weight", 2.0) * (bbox_loss / n_matched)
        B, C, h, w = out["class_logits"].shape
        pred_boxes = out["bbox_deltas"].permute(0, 2, 3, 1).reshape(B, h * w, 4)
        n_tgt = (gt_labels != -1).sum(dim=1)
        mets = val_metrics.detection_metrics(pred_boxes, box_xyxy_to_cxcywh(gt_boxes.float()), n_tgt, rows, cols, count)
        return loss, mets

    @torch.no_grad()
    def _evaluate_segmentation_batch(self, batch):
        """train_bdd100k_ddp.py:297-330: CE(ignore 255), pixel accuracy, mean IoU over the classes present."""
        logits = self.model(batch["image"].to(self.device))
        masks = batch["mask"].to(self.device)
        return hops.CrossEntropy2d.apply(logits, masks, 255), val_metrics.segmentation_metrics(logits, masks, 255)

    @torch.no_grad()
    def validate(self, epoch):
        """Mean validation loss over the loader (the reference's return value, train_bdd100k_ddp.py:332-397); the epoch's metrics
        (avg_iou / recall_0.5, or pixel_acc / mean_iou: means over batches, as the reference aggregates them) are left in
        ``self.last_val_metrics``.  Accumulated on the device, read once."""
        self.model.eval()
        total = torch.zeros((), device=self.device)
        agg = {}
        n = 0
        for batch in self.val_loader:
            loss, mets = self._evaluate_detection_batch(batch) if self.task == "detection" else self._evaluate_segmentation_batch(batch)
            total += loss
            for k, v in mets.items():
                agg[k] = agg.get(k, 0) + v.float()
            n += 1
        keys = sorted(agg)
        t = torch.stack([total, torch.tensor(float(n), device=self.device)] + [agg[k] for k in keys])
        if dist.is_initialized():
            dist.all_reduce(t)
        vals = t.tolist()
        denom = max(1.0, vals[1])
        self.last_val_metrics = {k: vals[2 + i] / denom for i, k in enumerate(keys)}
        return vals[0] / denom

    def save_best(self, epoch, val_loss):
        if dist.is_initialized() and dist.get_rank() != 0:
            return
        ckpt_dir = Path(f"models/checkpoints/bdd100k_{self.task}_expert/{self.config['run_name']}")
        ckpt_dir.mkdir(parents=True, exist_ok=True)
        torch.save({"epoch": epoch, "model_state_dict": self.core.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
                    "scheduler_state_dict": self.scheduler.state_dict(), "best_val_loss": val_loss, "config": self.config},
                   ckpt_dir / "best.pth")

    def train(self):
        for epoch in range(self.config["epochs"]):
            tr = self.train_epoch(epoch)
            va = self.validate(epoch)
            if (not dist.is_initialized()) or dist.get_rank() == 0:
                mets = " ".join(f"{k} {v:.4f}" for k, v in getattr(self, "last_val_metrics", {}).items())
                print(f"Epoch {epoch + 1}/{self.config['epochs']}: train {tr:.4f} val {va:.4f} {mets} skipped {int(self.optimizer.skipped)}")
            if va < self.best_val_loss:
                self.best_val_loss = va
                self.save_best(epoch, va)
            if dist.is_initialized():
                dist.barrier()


def main(argv=None):
    p = argparse.ArgumentParser(description="Train BDD100K Expert Models (MI355X HIP path)")
    p.add_argument("--task", type=str, required=True, choices=["detection", "drivable", "segmentation"])
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
    p.add_argument("--bbox_loss_weight", type=float, default=2.0)
    p.add_argument("--imagenet_norm", action="store_true")
    p.add_argument("--resume_from", type=str, default="")
    p.add_argument("--resume_mode", type=str, choices=["model", "full"], default="model")
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--synthetic", action="store_true", help="synthetic BDD-shaped batches (no dataset offline)")
    p.add_argument("--synthetic_steps", type=int, default=20)
    p.add_argument("--pretrained_backbone", action="store_true", help="needs AUTOMOE_RESNET18_WEIGHTS (reference default fetches)")
    p.add_argument("--precision", choices=["fp16", "fp32"], default="fp16")
    args = p.parse_args(argv)
    from .. import runtime
    runtime.set_compute_dtype(torch.float16 if args.precision == "fp16" else torch.float32)
    if args.imagenet_norm:  # applied to uint8 frames inside the boundary layout kernel (the reference: T.Normalize in the loader)
        runtime.set_input_normalization(runtime.IMAGENET_MEAN, runtime.IMAGENET_STD)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        dist.init_process_group(backend="nccl", init_method="env://")  # RCCL
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device(args.device)
    config = vars(args)
    cls = {"detection": BDDDetectionExpert, "drivable": BDDDrivableExpert, "segmentation": BDDSegmentationExpert}[args.task]
    model = cls(pretrained_backbone=args.pretrained_backbone).to(device)
    if not args.synthetic:
        raise SystemExit("BDD100K is not available offline; run with --synthetic")
    rank = int(os.environ.get("RANK", "0"))
    if args.task == "detection":
        batch = synthetic.bdd_detection_batch(args.batch_size, device=device, seed=rank)
    else:
        batch = synthetic.bdd_drivable_batch(args.batch_size, num_classes=model.num_classes, device=device, seed=rank)
    loader = synthetic.SyntheticLoader(batch, args.synthetic_steps)
    wrapped = DataParallel(model) if dist.is_initialized() else model
    trainer = BDDTrainer(args.task, wrapped, loader, loader, device, config)
    if args.resume_from:
        ck = torch.load(args.resume_from, map_location=device, weights_only=True)
        trainer.core.load_state_dict(ck.get("model_state_dict", ck), strict=True)
        if args.resume_mode == "full" and isinstance(ck, dict):
            trainer.load_training_state(ck)
    trainer.train()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
