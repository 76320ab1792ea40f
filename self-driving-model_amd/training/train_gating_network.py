"""AutoMoE gating-stage trainer -- drop-in for training/train_gating_network.py (compute_gating_losses :21-74,
train_one_epoch :76-117, validate :119-158, save/load_checkpoint :160-191, main :193-343).

Differences in HOW: the step glue is FusedAdamW (clip 1.0 folded in, norm on device) + GradBucketReducer
(RCCL all-reduce overlapped with backward) instead of torch DDP + clip_grad_norm_ + AdamW, there is no
per-step .item() sync (losses are accumulated on the device and read once per epoch), and `--synthetic`
replaces the CARLA dataset (absent offline) with batches of the same dict layout.
"""
import argparse
import json
import os
from typing import Dict

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (kept for API parity with the reference module namespace)

from .. import runtime
from ..models.automoe import create_automoe_model
from . import synthetic
from .ddp import DataParallel, GradBucketReducer, StepStream, capture_error_mode, capture_step, detached, quiesce_collectives
from .optim import FusedAdamW


def _l1(a, b):
    return (a - b).abs().mean()


def compute_gating_losses(pred: Dict[str, torch.Tensor], target_wp: torch.Tensor, target_spd: torch.Tensor,
                          config: Dict) -> Dict[str, torch.Tensor]:
    """ADE / FDE / speed L1 + smoothness + load-balance MSE + negative entropy; weights 1, 2, .2, .1, .01, .001."""
    wp = pred["waypoints"]
    ade = _l1(wp, target_wp)
    fde = _l1(wp[:, -1, :], target_wp[:, -1, :])
    pred_spd = pred.get("speed_seq", pred.get("speed"))
    if pred_spd is not None and pred_spd.dim() == 2 and target_spd.dim() == 2 and pred_spd.size(1) == target_spd.size(1):
        speed_loss = _l1(pred_spd, target_spd)
    else:
        pred_last = pred.get("speed")
        if pred_last is not None and pred_last.dim() == 2 and pred_last.size(1) == 1:
            speed_loss = _l1(pred_last, target_spd[:, -1:].contiguous())
        else:
            speed_loss = torch.zeros((), device=target_spd.device)
    d = wp[:, 1:, :] - wp[:, :-1, :]
    smoothness_loss = _l1(d[:, 1:, :], d[:, :-1, :])
    w = pred["expert_weights"]
    if config.get("use_load_balancing", True):
        usage = w.mean(dim=0)
        load_balancing_loss = ((usage - 1.0 / usage.size(0)) ** 2).mean()
    else:
        load_balancing_loss = torch.tensor(0.0, device=w.device)
    if config.get("use_entropy_loss", True):
        entropy_loss = (w * torch.log(w + 1e-8)).sum(dim=1).mean()  # = -entropy
    else:
        entropy_loss = torch.tensor(0.0, device=w.device)
    total = (config.get("ade_weight", 1.0) * ade + config.get("fde_weight", 2.0) * fde
             + config.get("speed_weight", 0.2) * speed_loss + config.get("smoothness_weight", 0.1) * smoothness_loss
             + config.get("load_balancing_weight", 0.01) * load_balancing_loss
             + config.get("entropy_weight", 0.001) * entropy_loss)
    return {"total_loss": total, "ade": ade, "fde": fde, "speed": speed_loss, "smoothness": smoothness_loss,
            "load_balancing": load_balancing_loss, "entropy": entropy_loss}


def fused_gating_losses(pred: Dict[str, torch.Tensor], target_wp: torch.Tensor, target_spd: torch.Tensor,
                        config: Dict) -> Dict[str, torch.Tensor]:
    """compute_gating_losses as one HIP launch (forward values + gradient): same dictionary, same arithmetic in fp32."""
    from ..hip import ops as hops
    wp = pred["waypoints"]
    pred_spd = pred.get("speed_seq", pred.get("speed"))
    spd = tspd = None
    if pred_spd is not None and pred_spd.dim() == 2 and target_spd.dim() == 2 and pred_spd.size(1) == target_spd.size(1):
        spd, tspd = pred_spd, target_spd
    else:
        pred_last = pred.get("speed")
        if pred_last is not None and pred_last.dim() == 2 and pred_last.size(1) == 1:
            spd, tspd = pred_last, target_spd[:, -1:]
    coef = [config.get("ade_weight", 1.0), config.get("fde_weight", 2.0), config.get("speed_weight", 0.2),
            config.get("smoothness_weight", 0.1), config.get("load_balancing_weight", 0.01), config.get("entropy_weight", 0.001)]
    total, v = hops.GatingLosses.apply(wp, target_wp, spd, tspd, pred["expert_weights"], coef,
                                       bool(config.get("use_load_balancing", True)), bool(config.get("use_entropy_loss", True)))
    return {"total_loss": total, "ade": v[0], "fde": v[1], "speed": v[2], "smoothness": v[3], "load_balancing": v[4],
            "entropy": v[5]}


class GatingTrainStep:
    """One optimisation step of the gating stage: zero_grad -> forward -> losses -> backward (+ overlapped
    all-reduce) -> clip 1.0 + AdamW.  Holds the optimizer / reducer pair so callers (trainer, bench) share it.

    With `use_graph` (default: env AUTOMOE_HIPGRAPH, on) the launch-bound part of the step -- zero_grad, forward,
    losses, backward: ~900 small launches -- is captured once into a hipGraph (torch.cuda.CUDAGraph over our kernels,
    which run on the capturing stream) and replayed; the gradient exchange and the fused optimizer stay outside the
    graph so the learning rate / step count remain host-driven.  With RCCL ("nccl") the per-bucket all-reduces are part
    of the captured step (side stream, beside the rest of backward: training/ddp.py); a backend that cannot be captured
    ("gloo" in tests) gets one collective after the replay instead.  A batch whose tensor shapes differ from the captured
    ones (the reference's gating loader has no drop_last: training/train_gating_network.py:259-267) runs eagerly with the
    same collectives."""

    def __init__(self, model: nn.Module, config: Dict, bucket_mb: int = None, use_graph=None):
        self.model = model
        self.core = model.module if hasattr(model, "module") else model
        self.config = config
        if hasattr(self.core, "fuse_expert_pooling"):
            self.core.fuse_expert_pooling = bool(config.get("fuse_expert_pooling", True))  # the step never reads expert_outputs
        params = [p for p in self.core.parameters() if p.requires_grad]
        self.optimizer = FusedAdamW(params, lr=config.get("learning_rate", 1e-4), weight_decay=config.get("weight_decay", 1e-4),
                                    max_norm=1.0)
        self.optimizer.attach_conv_packs(self.core.modules())
        self.reducer = GradBucketReducer(self.optimizer._params, self.optimizer._offsets, self.optimizer.flat_g,
                                         bucket_bytes=None if bucket_mb is None else bucket_mb << 20,
                                         broadcast_from=self.optimizer.flat_p)  # None: AUTOMOE_BUCKET_MB or 25 MB
        self.optimizer.grad_divisor = float(self.reducer.world)
        if use_graph is None:
            use_graph = os.environ.get("AUTOMOE_HIPGRAPH", "1") != "0"
        self.use_graph = bool(use_graph)
        self._graph = None
        self._static_batch = None
        self._static_losses = None
        self._reduce_in_graph = False   # the captured step holds its bucket all-reduces (RCCL)
        self._eager_steps = 0
        self._step_stream = StepStream(self.optimizer.flat_p.device)  # every step, eager or captured, runs on this stream
        # Expert prefetch (frozen experts only): the experts' forward is its own hipGraph on its own stream and runs for the
        # NEXT batch while the rest of this step -- gating / policy forward, backward, all-reduce, optimizer: kernels that
        # are HBM-, atomics- or latency-bound -- is still on the GPU beside the experts' MFMA-bound convolutions.  Frozen
        # experts do not depend on the update, so the training trajectory is unchanged.
        self.prefetch_experts = os.environ.get("AUTOMOE_PREFETCH_EXPERTS", "1") != "0"
        self._graph_experts = None      # graph A: forward_experts(static expert batch)
        self._expert_stream = None
        self._expert_batch = None       # graph A's static input ({"image": ...})
        self._cache_a = self._cache_b = None  # expert results: as graph A writes them / the copy graph B reads
        self._ev_experts = self._ev_copied = None
        self._experts_pending = False   # graph A already launched for the batch of the next call

    def _fwd_bwd(self, batch, expert_cache=None):
        self.optimizer.zero_grad()
        # Linear / LayerNorm / BatchNorm parameter gradients go straight into the flat buffer; their kernels' call sites tell
        # the bucketed all-reduce when one is complete (runtime.grad_ready), autograd's hooks cover the rest
        runtime.set_direct_grads(True)
        try:
            pred = self.model(batch) if expert_cache is None else self.model(batch, expert_cache=expert_cache)
            losses = fused_gating_losses(pred, batch["waypoints"], batch["speed"], self.config)
            losses["total_loss"].backward()
        finally:
            runtime.set_direct_grads(False)
        return losses

    def _capture(self, batch):
        self._static_batch = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        mode = capture_error_mode()  # thread_local: a live process group's watchdog polls events from its own thread (training/ddp.py)
        cache_b = None
        try:
            quiesce_collectives(self.reducer)  # (the expert graph below is captured outside capture_step)
            if (self.prefetch_experts and hasattr(self.core, "forward_experts") and self.core.fuse_expert_pooling
                    and self.core.experts_frozen() and batch["image"].is_cuda):
                self._expert_stream = torch.cuda.Stream(device=batch["image"].device)
                self._expert_batch = {"image": self._static_batch["image"].clone()}
                ga = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, capture_error_mode=mode):
                    cache_a = self.core.forward_experts(self._expert_batch)
                # graph B reads its own copy of the experts' results (tensors or dicts of tensors), taken right before its replay
                flat_a, flat_b = [], []

                def mirror(v):
                    if isinstance(v, torch.Tensor):
                        flat_a.append(v)
                        flat_b.append(torch.empty_like(v))
                        return flat_b[-1]
                    if isinstance(v, dict):
                        return {k: mirror(x) for k, x in v.items()}
                    if isinstance(v, (list, tuple)):
                        return type(v)(mirror(x) for x in v)
                    return v

                cache_b = {"pend": [(mirror(t), pooled) for t, pooled in cache_a["pend"]], "outs": [mirror(o) for o in cache_a["outs"]]}
                self._cache_a, self._cache_b = flat_a, flat_b
                # the per-step hand-over copies graph A's results into graph B's mirrors: one multi-tensor launch per dtype
                # (torch._foreach_copy_ takes its one-kernel route only for lists of ONE dtype; the mixed f16 / fp32 list
                # fell back to a hipMemcpy per tensor: ~85 copyBuffer launches per step in round 2's trace)
                groups = {}
                for a_, b_ in zip(flat_a, flat_b):
                    groups.setdefault((a_.dtype, a_.is_contiguous()), ([], []))
                    groups[(a_.dtype, a_.is_contiguous())][0].append(a_)
                    groups[(a_.dtype, a_.is_contiguous())][1].append(b_)
                self._cache_groups = list(groups.values())
                self._ev_experts, self._ev_copied = torch.cuda.Event(), torch.cuda.Event()
                self._graph_experts = ga
                # one-time costs (a graph's first launch on a new stream, the copy kernels' module load: ~10 ms each) are paid
                # here, not in the caller's first step.  The experts' BatchNorm buffers (all graph A changes) are put back.
                bufs = [b for e in self.core.experts for b in e.buffers()]
                saved = [b.clone() for b in bufs]
                torch.cuda.synchronize()
                with torch.cuda.stream(self._expert_stream):
                    ga.replay()
                for ga_, gb_ in self._cache_groups:
                    torch._foreach_copy_(gb_, ga_)
                torch.cuda.synchronize()
                for b, v in zip(bufs, saved):
                    b.copy_(v)
        except Exception as e:  # noqa: BLE001  (the prefetch is an optimisation: go on without it, loudly)
            import traceback
            import warnings
            warnings.warn(f"hipGraph capture of the expert prefetch failed ({e!r}); continuing without it\n"
                          + "".join(traceback.format_exc().splitlines(True)[-14:]))
            self._graph_experts, cache_b = None, None
            torch.cuda.synchronize()

        def capture(in_graph):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._step_stream.stream, capture_error_mode=mode):
                losses = detached(self._fwd_bwd(self._static_batch, cache_b))  # same storage, no graph kept alive
                if in_graph:
                    self.reducer.finish()  # records the joins: the replay ends with every bucket reduced
            return g, losses

        # (ranks agree on one mode inside capture_step: graph and eager ranks would otherwise issue different collectives)
        self._graph, self._static_losses, self._reduce_in_graph = capture_step(self.reducer, capture, "gating train step")
        if self._graph is None:
            self.use_graph = False
            self._graph_experts = None

    @property
    def input_buffers(self):
        """The graph's static input tensors once the step is captured (None before): a loader that writes its host-to-device
        copies straight into them (or passes them back to ``__call__``) saves the per-step device-to-device copy."""
        return self._static_batch if self._graph is not None else None

    def _fits_graph(self, batch) -> bool:
        """`batch` has the captured step's tensor shapes / dtypes (a ragged last batch does not)."""
        return all(isinstance(batch.get(k), torch.Tensor) and batch[k].shape == v.shape and batch[k].dtype == v.dtype
                   for k, v in self._static_batch.items() if isinstance(v, torch.Tensor))

    def _launch_experts(self, batch) -> bool:
        """Graph A for `batch` on the expert stream (after the previous step's copy of its results has been taken).
        False (nothing launched) when the batch's image does not have the captured shape."""
        if batch is not True:  # True: the caller's loader already wrote the batch into expert_input_buffers
            img = batch.get("image") if isinstance(batch, dict) else None
            ref = self._expert_batch["image"]
            if not isinstance(img, torch.Tensor) or img.shape != ref.shape or img.dtype != ref.dtype:
                return False
        st = self._expert_stream
        st.wait_event(self._ev_copied)  # (never recorded yet: no-op)
        with torch.cuda.stream(st):
            if batch is not True and batch["image"].data_ptr() != self._expert_batch["image"].data_ptr():
                self._expert_batch["image"].copy_(batch["image"], non_blocking=True)
            self._graph_experts.replay()
            runtime.bump_stats_epoch()  # frozen experts still update their BatchNorm running statistics (train mode)
            self._ev_experts.record(st)
        return True

    @property
    def expert_input_buffers(self):
        """Graph A's static input ({"image": ...}) when the expert prefetch is active (else None): where a loader puts the
        NEXT batch's image, to be passed as ``next_batch`` without a device-to-device copy."""
        return self._expert_batch if self._graph_experts is not None and self._graph is not None else None

    def _eager_step_beside_graph(self, batch):
        """A batch the captured step cannot take, after the capture: the same step eagerly, exchanging gradients with the
        collectives the replaying ranks issue (per bucket when they are part of the graph, else the single all-reduce)."""
        if self._graph_experts is not None:
            main = torch.cuda.current_stream()
            main.wait_event(self._ev_experts)  # the experts' BatchNorm buffers: a pending prefetch finishes first
        self.reducer.paused = self.reducer.enabled and not self._reduce_in_graph
        self.reducer.reset()
        losses = self._fwd_bwd(batch)
        if self._reduce_in_graph:
            self.reducer.finish()
        else:
            self.reducer.reduce_all()
        if self._graph_experts is not None:
            self._ev_copied.record(torch.cuda.current_stream())  # the next prefetch waits for this step's expert forward
        return losses

    def __call__(self, batch: Dict[str, torch.Tensor], next_batch: Dict[str, torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """One step on `batch`.  `next_batch` (optional): the batch of the NEXT call -- a dict with its "image", or True when
        the caller's loader writes the next image straight into ``expert_input_buffers`` (before this step is captured: the
        first batch, which the buffer is initialised with).  With frozen experts and a captured step its expert forward is
        launched now and overlaps this step's backward / optimizer; the next call must then be made with exactly that batch."""
        with self._step_stream:
            return self._step(batch, next_batch)

    def _step(self, batch, next_batch):
        if self.use_graph and self._graph is None and self._eager_steps >= 2 and self.model.training:
            self._capture(batch)
        if self._graph is not None and self._fits_graph(batch):
            for k, v in batch.items():
                if isinstance(v, torch.Tensor) and v.data_ptr() != self._static_batch[k].data_ptr():
                    self._static_batch[k].copy_(v, non_blocking=True)
            if self._graph_experts is not None:
                if not self._experts_pending:
                    self._launch_experts(batch)
                main = torch.cuda.current_stream()
                main.wait_event(self._ev_experts)
                for ga_, gb_ in self._cache_groups:  # graph A may now overwrite its results
                    torch._foreach_copy_(gb_, ga_)
                self._ev_copied.record(main)
                self._experts_pending = False
            self._graph.replay()
            runtime.bump_stats_epoch()
            losses = self._static_losses
            if self._graph_experts is not None and next_batch is not None:
                self._experts_pending = self._launch_experts(next_batch)
            if not self._reduce_in_graph:
                self.reducer.reduce_all()
        elif self._graph is not None:
            # ragged batch (a prefetch launched for it would have been refused by _launch_experts: nothing is pending)
            self._experts_pending = False
            losses = self._eager_step_beside_graph(batch)
            if self._graph_experts is not None and next_batch is not None:
                self._experts_pending = self._launch_experts(next_batch)
        else:
            losses = self._fwd_bwd(batch)
            self.reducer.finish()
            self._eager_steps += 1
        self.optimizer.step()
        return detached(losses)


def train_one_epoch(model, loader, optimizer, device, epoch_idx, epochs, rank, config, step: GatingTrainStep = None) -> float:
    model.train()
    step = step or GatingTrainStep(model, config)
    total = torch.zeros((), device=device)
    n = 0
    to_dev = lambda b: {k: v.to(device) if isinstance(v, torch.Tensor) else v for k, v in b.items()}  # noqa: E731
    it = iter(loader)
    nxt = next(it, None)
    nxt = to_dev(nxt) if nxt is not None else None
    while nxt is not None:
        batch = nxt
        nxt = next(it, None)
        nxt = to_dev(nxt) if nxt is not None else None
        losses = step(batch, next_batch=nxt)  # one batch of lookahead: the frozen experts run on it during this step's backward
        total += losses["total_loss"].detach()
        n += 1
    return float(total.item()) / max(1, n)


@torch.no_grad()
def validate(model, loader, device, epoch_idx, epochs, rank, world_size, config) -> float:
    model.eval()
    total = torch.zeros((), device=device)
    n = 0
    for batch in loader:
        batch = {k: v.to(device) if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
        pred = model(batch)
        total += compute_gating_losses(pred, batch["waypoints"], batch["speed"], config)["total_loss"]
        n += 1
    t = torch.stack([total, torch.tensor(float(n), device=device)])
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0].item()) / max(1, int(t[1].item()))


def save_checkpoint(model, optimizer, epoch, loss, save_path, rank):
    if rank == 0:
        torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                    "loss": loss}, save_path)
        print(f"Saved checkpoint to {save_path}")


def load_checkpoint(model, optimizer, checkpoint_path, device) -> int:
    if os.path.exists(checkpoint_path):
        ck = torch.load(checkpoint_path, map_location=device, weights_only=True)
        model.load_state_dict(ck["model_state_dict"])
        optimizer.load_state_dict(ck["optimizer_state_dict"])
        print(f"Loaded checkpoint from {checkpoint_path}, starting from epoch {ck['epoch'] + 1}")
        return ck["epoch"] + 1
    print(f"No checkpoint found at {checkpoint_path}, starting from epoch 0")
    return 0


def main(argv=None):
    parser = argparse.ArgumentParser(description="Train AutoMoE gating network (MI355X HIP path)")
    parser.add_argument("--config", type=str, required=True)
    parser.add_argument("--model_config", type=str, default=os.path.join(os.path.dirname(__file__), "..", "models", "configs", "automoe", "model_config.json"))
    parser.add_argument("--data_root", type=str, default="")
    parser.add_argument("--checkpoint_dir", type=str, default="models/checkpoints/gating")
    parser.add_argument("--resume", type=str, default="")
    parser.add_argument("--expert_checkpoints", nargs="+", default=[])
    parser.add_argument("--local_rank", type=int, default=0)
    parser.add_argument("--world_size", type=int, default=1)
    parser.add_argument("--synthetic", action="store_true", help="synthetic CARLA-shaped batches (no dataset offline)")
    parser.add_argument("--synthetic_steps", type=int, default=20)
    parser.add_argument("--precision", choices=["fp16", "fp32"], default="fp16")
    args = parser.parse_args(argv)
    from .. import runtime
    runtime.set_compute_dtype(torch.float16 if args.precision == "fp16" else torch.float32)
    with open(args.config) as f:
        config = json.load(f)
    with open(args.model_config) as f:
        model_config = json.load(f)
    world = int(os.environ.get("WORLD_SIZE", str(args.world_size)))
    if world > 1:
        args.local_rank = int(os.environ.get("LOCAL_RANK", str(args.local_rank)))
        args.world_size = world
        dist.init_process_group(backend="nccl", init_method="env://")  # "nccl" is RCCL on ROCm
        torch.cuda.set_device(args.local_rank)
    device = torch.device(f"cuda:{args.local_rank}")
    model = create_automoe_model(model_config, device)
    if args.expert_checkpoints:
        model.load_expert_checkpoints(args.expert_checkpoints)
    model.freeze_experts()
    wrapped = DataParallel(model) if world > 1 else model
    if not args.synthetic:
        raise SystemExit("the CARLA dataset is not available offline; run with --synthetic")
    H, W = config.get("image_size", [720, 1280])
    rank = int(os.environ.get("RANK", "0"))
    batch = synthetic.carla_sequence_batch(config.get("batch_size", 8), H, W, config.get("sequence_length", 10), device, seed=rank)
    loader = synthetic.SyntheticLoader(batch, args.synthetic_steps)
    step = GatingTrainStep(wrapped, config)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(step.optimizer, T_max=config.get("epochs", 100) * len(loader))
    start = load_checkpoint(wrapped, step.optimizer, args.resume, device) if args.resume else 0
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    best = float("inf")
    for epoch in range(start, config.get("epochs", 100)):
        tr = train_one_epoch(wrapped, loader, step.optimizer, device, epoch, config.get("epochs", 100), args.local_rank, config, step)
        va = validate(wrapped, loader, device, epoch, config.get("epochs", 100), args.local_rank, world, config)
        sched.step()  # once per epoch, as the reference (:314)
        if rank == 0:
            print(f"Epoch {epoch + 1}: Train Loss: {tr:.4f}, Val Loss: {va:.4f}, skipped steps: {int(step.optimizer.skipped)}")
        if va < best:
            best = va
            save_checkpoint(wrapped, step.optimizer, epoch, va, os.path.join(args.checkpoint_dir, "best.pth"), rank)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
