"""FusedAdamW: clip_grad_norm_(max_norm) + torch.optim.AdamW.step() as two HIP launches over flat
fp32 buffers (csrc/optim.hip), with the gradient norm kept on the device (the reference's step glue at
training/train_bdd100k_ddp.py:98-99 and training/train_gating_network.py:103-105).

Parameters are re-pointed at views of one flat buffer (p.data) and their .grad at views of one flat
gradient buffer, so zero_grad is one memset, the data-parallel all-reduce works on contiguous bucket
slices (training/ddp.py) and the update is one pass.  Subclasses torch.optim.Optimizer so torch LR
schedulers (CosineAnnealingLR) drive param_groups[0]['lr'] as in the reference.
"""
from __future__ import annotations

from typing import Iterable, List

import os

import torch

from .. import runtime
from ..hip import conv as _conv
from ..hip import lib as _lib
from ..hip.conv import ptr, require_hip, stream


GROUP_CONV_PACKS = os.environ.get("AUTOMOE_GROUP_PACKS", "1") != "0"  # tests flip this to compare with the per-layer re-pack on first use


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_norm: float = 0.0):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FusedAdamW got no trainable parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = float(max_norm)
        self._params: List[torch.nn.Parameter] = params
        dev = params[0].device
        require_hip(params[0], "parameters")
        sizes = [p.numel() for p in params]
        self._offsets, off = [], 0
        for n in sizes:
            self._offsets.append(off)
            off += (n + 3) // 4 * 4  # keep every view 16-byte aligned
        self.numel = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(params, self._offsets):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
        self.norm_sq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        self.step_count = 0
        self.grad_divisor = 1.0  # world size when gradients arrive summed (training/ddp.py)
        self.pack_group = None   # hip.conv.PackGroup: the conv layers' packed operands, rebuilt in one launch per step

    def attach_conv_packs(self, modules):
        """Conv layers among `modules` whose weights this optimizer owns get their packed operands (forward / input-gradient
        layouts) rebuilt by ONE launch at the start of every step -- zero_grad() is the first call of a step -- instead of one
        launch per layer on first use."""
        from ..hip.conv import PackGroup
        self.pack_group = PackGroup(modules, self.flat_p)
        return self.pack_group

    def zero_grad(self, set_to_none: bool = False):
        if self.pack_group is not None and GROUP_CONV_PACKS:
            from .. import runtime
            self.pack_group.refresh(runtime.compute_dtype())
        self.flat_g.zero_()
        for p, o in zip(self._params, self._offsets):  # autograd may have replaced a .grad view; restore it
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def _gather_stray_grads(self):
        for p, o in zip(self._params, self._offsets):
            if p.grad is not None and p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                self.flat_g[o:o + p.numel()].view(p.shape).add_(p.grad)
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None):
        L = _lib.get()
        _conv.assert_residual_handoff_consumed()
        self._gather_stray_grads()
        self.step_count += 1
        group = self.param_groups[0]
        if self.grad_divisor != 1.0:
            L.am_scale_inplace(ptr(self.flat_g), self.numel, 1.0 / self.grad_divisor, None, stream())
        self.norm_sq.zero_()
        L.am_sumsq_accumulate(ptr(self.flat_g), self.numel, ptr(self.norm_sq), stream())
        b1, b2 = group["betas"]
        L.am_adamw_step(ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.numel,
                        float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                        self.step_count, self.max_norm, ptr(self.norm_sq), ptr(self.skipped), stream())
        runtime.bump_weight_epoch()

    def grad_norm(self) -> torch.Tensor:
        """Pre-clip global gradient norm of the last step (device scalar; reading it synchronises)."""
        return self.norm_sq.sqrt().float()

    def state_dict(self):
        """torch.optim.AdamW's layout -- state[i] = {'step', 'exp_avg', 'exp_avg_sq'} per parameter index, sliced out of the
        flat moment buffers -- so a checkpoint written here resumes under the reference's torch AdamW
        (train_bdd100k_ddp.py:536-545 `--resume_mode full`) and the reverse."""
        sd = super().state_dict()
        if self.step_count > 0:
            step = torch.tensor(float(self.step_count))
            sd["state"] = {i: {"step": step.clone(), "exp_avg": self.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                               "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone()}
                           for i, (p, o) in enumerate(zip(self._params, self._offsets))}
        return sd

    def load_state_dict(self, state_dict):
        fused = state_dict.get("fused")  # round-1 checkpoints of this build: the flat buffers as they are
        state = state_dict.get("state") or {}
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        if fused is not None:
            self.exp_avg.copy_(fused["exp_avg"])
            self.exp_avg_sq.copy_(fused["exp_avg_sq"])
            self.step_count = int(fused["step"])
            return
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count = 0
        for i, (p, o) in enumerate(zip(self._params, self._offsets)):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue  # torch creates a parameter's state lazily: no entry = never stepped
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {i}: exp_avg shape {tuple(st['exp_avg'].shape)} != parameter shape {tuple(p.shape)}")
            self.exp_avg[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
            self.step_count = max(self.step_count, int(float(st["step"])))  # one bias-correction count for the fused pass
