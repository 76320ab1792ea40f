"""CARLA trajectory-policy trainer -- drop-in for training/train_carla_policy.py (compute_losses :22-30, train_one_epoch
:33-56, validate :59-82, train :85-127, main :141-208) on the MI355X HIP path.

`compute_losses` keeps the reference's arithmetic in torch ops (any device: used by validation and as the test oracle);
`fused_losses` is the same objective + gradient as one HIP launch (am_gating_losses with the two gating terms off);
`PolicyTrainStep` = zero_grad -> forward -> loss -> backward (+ gradient all-reduce) -> clip 1.0 + AdamW(lr, wd 1e-4), the
forward/backward part captured into a hipGraph like the gating stage.
"""
import argparse
import json
import os
from pathlib import Path
from typing import Dict, Optional

import torch
import torch.distributed as dist

from .. import runtime
from ..models.policy.trajectory_head import TrajectoryPolicy
from . import synthetic
from .ddp import DataParallel, GradBucketReducer, StepStream, capture_error_mode, capture_step, detached
from .optim import FusedAdamW


def compute_losses(pred: Dict[str, torch.Tensor], target_wp: torch.Tensor, target_spd: torch.Tensor) -> Dict[str, torch.Tensor]:
    """train_carla_policy.py:22-30: ADE + 2 FDE + 0.2 speed L1 + 0.1 smoothness (L1 of second differences)."""
    l1 = lambda a, b: (a - b).abs().mean()
    wp = pred["waypoints"]
    ade = l1(wp, target_wp)
    fde = l1(wp[:, -1, :], target_wp[:, -1, :])
    l_spd = l1(pred["speed"], target_spd)
    d = wp[:, 1:, :] - wp[:, :-1, :]
    l_smooth = l1(d[:, 1:, :], d[:, :-1, :])
    return {"loss": ade + 2.0 * fde + 0.2 * l_spd + 0.1 * l_smooth, "ade": ade, "fde": fde, "speed": l_spd, "smooth": l_smooth}


def fused_losses(pred: Dict[str, torch.Tensor], target_wp: torch.Tensor, target_spd: torch.Tensor) -> Dict[str, torch.Tensor]:
    from ..hip import ops as hops
    wp = pred["waypoints"]
    ones = torch.ones((wp.shape[0], 1), dtype=torch.float32, device=wp.device)  # unused expert-weight slot of the kernel
    total, v = hops.GatingLosses.apply(wp, target_wp, pred["speed"], target_spd, ones, [1.0, 2.0, 0.2, 0.1, 0.0, 0.0], False, False)
    return {"loss": total, "ade": v[0], "fde": v[1], "speed": v[2], "smooth": v[3]}


class PolicyTrainStep:
    def __init__(self, model, lr: float = 3e-4, weight_decay: float = 1e-4, use_graph: Optional[bool] = None):
        self.model = model
        self.core = model.module if hasattr(model, "module") else model
        self.optimizer = FusedAdamW(self.core.parameters(), lr=lr, weight_decay=weight_decay, max_norm=1.0)
        self.optimizer.attach_conv_packs(self.core.modules())
        self.reducer = GradBucketReducer(self.optimizer._params, self.optimizer._offsets, self.optimizer.flat_g,
                                         broadcast_from=self.optimizer.flat_p)
        self.optimizer.grad_divisor = float(self.reducer.world)
        self.use_graph = (os.environ.get("AUTOMOE_HIPGRAPH", "1") != "0") if use_graph is None else bool(use_graph)
        self._graph = self._static = self._static_losses = None
        self._reduce_in_graph = False
        self._eager_steps = 0
        self._step_stream = StepStream(self.optimizer.flat_p.device)  # training/ddp.py StepStream

    def _fwd_bwd(self, batch):
        self.optimizer.zero_grad()
        runtime.set_direct_grads(True)  # the direct kernels report to the reducer through runtime.grad_ready()
        try:
            losses = fused_losses(self.model(batch["image"], batch.get("context")), batch["waypoints"], batch["speed"])
            losses["loss"].backward()
        finally:
            runtime.set_direct_grads(False)
        return losses

    def _capture(self, batch):
        self._static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        mode = capture_error_mode()  # thread_local: a live process group's watchdog polls events from its own thread (training/ddp.py)

        def capture(in_graph):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._step_stream.stream, capture_error_mode=mode):
                losses = detached(self._fwd_bwd(self._static))
                if in_graph:
                    self.reducer.finish()
            return g, losses

        self._graph, self._static_losses, self._reduce_in_graph = capture_step(self.reducer, capture, "policy train step")
        if self._graph is None:
            self.use_graph = False

    def _fits_graph(self, batch) -> bool:
        return all(isinstance(batch.get(k), torch.Tensor) and batch[k].shape == v.shape and batch[k].dtype == v.dtype
                   for k, v in self._static.items() if isinstance(v, torch.Tensor))

    def __call__(self, batch):
        with self._step_stream:
            return self._step(batch)

    def _step(self, batch):
        if self.use_graph and self._graph is None and self._eager_steps >= 2 and self.core.training:
            self._capture(batch)
        if self._graph is not None and self._fits_graph(batch):
            for k, v in batch.items():
                if isinstance(v, torch.Tensor) and v.data_ptr() != self._static[k].data_ptr():
                    self._static[k].copy_(v, non_blocking=True)
            self._graph.replay()
            runtime.bump_stats_epoch()
            losses = self._static_losses
            if not self._reduce_in_graph:
                self.reducer.reduce_all()
        elif self._graph is not None:  # ragged batch after the capture: eager, same collectives as the replaying ranks
            self.reducer.reset()
            losses = self._fwd_bwd(batch)
            if self._reduce_in_graph:
                self.reducer.finish()
            else:
                self.reducer.reduce_all()
        else:
            losses = self._fwd_bwd(batch)
            self.reducer.finish()
            self._eager_steps += 1
        self.optimizer.step()
        return detached(losses)


def train_one_epoch(model, loader, step: PolicyTrainStep, device, epoch_idx: int, epochs: int, rank: int) -> float:
    model.train()
    total = torch.zeros((), device=device)
    for batch in loader:
        batch = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        total += step(batch)["loss"].detach()
    return float(total.item()) / max(1, len(loader))


@torch.no_grad()
def validate(model, loader, device, epoch_idx: int, epochs: int, rank: int, world_size: int) -> float:
    model.eval()
    total = torch.zeros((), device=device)
    count = 0
    for batch in loader:
        pred = model(batch["image"].to(device), batch["context"].to(device) if batch.get("context") is not None else None)
        total += compute_losses(pred, batch["waypoints"].to(device), batch["speed"].to(device))["loss"]
        count += 1
    t = torch.stack([total, torch.tensor(float(count), device=device)])
    if dist.is_initialized():
        dist.all_reduce(t)
    return float(t[0].item()) / max(1.0, float(t[1].item()))


def train(model, train_loader, val_loader, step: PolicyTrainStep, device, epochs, rank, world_size, run_name, ckpt_root) -> float:
    best_val = float("inf")
    ckpt_dir = Path(ckpt_root) / run_name
    core = model.module if hasattr(model, "module") else model
    for epoch in range(epochs):
        tr = train_one_epoch(model, train_loader, step, device, epoch, epochs, rank)
        va = validate(model, val_loader, device, epoch, epochs, rank, world_size)
        if rank == 0:
            if va < best_val:
                best_val = va
                ckpt_dir.mkdir(parents=True, exist_ok=True)
                torch.save({"epoch": epoch + 1, "model_state_dict": core.state_dict(), "optimizer_state_dict": step.optimizer.state_dict(),
                            "best_val_loss": best_val, "horizon": getattr(core, "horizon", None)}, ckpt_dir / "best.pth")
            print(f"epoch {epoch + 1}/{epochs}: train {tr:.4f} | val {va:.4f} | best {best_val:.4f}")
    if rank == 0 and epochs > 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
        torch.save({"epoch": epochs, "model_state_dict": core.state_dict(), "optimizer_state_dict": step.optimizer.state_dict(),
                    "best_val_loss": best_val, "horizon": getattr(core, "horizon", None)}, ckpt_dir / "last.pth")
    return best_val


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--data_root", type=str, default="datasets/carla/preprocessed")
    p.add_argument("--epochs", type=int, default=0, help="0 means dry-run (no training)")
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--num_workers", type=int, default=4)
    p.add_argument("--horizon", type=int, default=8)
    p.add_argument("--lr", type=float, default=3e-4)
    p.add_argument("--no_context", action="store_true")
    p.add_argument("--run_name", type=str, default="carla_policy_ddp")
    p.add_argument("--ckpt_dir", type=str, default="models/checkpoints/carla_policy")
    p.add_argument("--synthetic", action="store_true", help="synthetic CARLA-shaped batches (no dataset offline)")
    p.add_argument("--synthetic_steps", type=int, default=20)
    p.add_argument("--context_dim", type=int, default=0, help="synthetic context width (the dataset's context vector size)")
    p.add_argument("--image_hw", type=int, nargs=2, default=[720, 1280])
    p.add_argument("--precision", choices=["fp16", "fp32"], default="fp16")
    args = p.parse_args(argv)
    runtime.set_compute_dtype(torch.float16 if args.precision == "fp16" else torch.float32)
    world_size, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world_size > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")  # RCCL
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if not args.synthetic:
        raise SystemExit("the CARLA dataset is not available offline; run with --synthetic")
    rank = int(os.environ.get("RANK", "0"))
    batch = synthetic.carla_sequence_batch(args.batch_size, args.image_hw[0], args.image_hw[1], args.horizon, device, seed=rank)
    batch = {"image": batch["image"], "waypoints": batch["waypoints"], "speed": batch["speed"]}
    ctx_dim = 0 if args.no_context else args.context_dim
    if ctx_dim:
        batch["context"] = torch.randn(args.batch_size, ctx_dim, device=device)
    loader = synthetic.SyntheticLoader(batch, args.synthetic_steps)
    model = TrajectoryPolicy(horizon=args.horizon, context_dim=ctx_dim).to(device)
    wrapped = DataParallel(model) if world_size > 1 else model
    if args.epochs <= 0:
        with torch.no_grad():
            out = wrapped(batch["image"][:1], batch["context"][:1] if ctx_dim else None)
        if rank == 0:
            print({k: tuple(v.shape) for k, v in out.items()})
        return
    if rank == 0:
        cfg_dir = Path("models/configs/carla_policy") / args.run_name
        cfg_dir.mkdir(parents=True, exist_ok=True)
        with open(cfg_dir / "config.json", "w") as f:
            json.dump(vars(args), f, indent=2)
    step = PolicyTrainStep(wrapped, lr=args.lr, weight_decay=1e-4)
    best = train(wrapped, loader, loader, step, device, args.epochs, rank, world_size, args.run_name, args.ckpt_dir)
    if rank == 0:
        print(f"training complete. best val {best:.4f}")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
