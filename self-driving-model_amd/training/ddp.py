"""Data-parallel gradient exchange for one process per GPU (the reference wraps its model in
torch DistributedDataParallel: training/train_bdd100k_ddp.py:497, training/train_gating_network.py:236).

One RCCL all-reduce (sum) per gradient bucket over xGMI, issued from autograd hooks as soon as every
gradient in the bucket has been accumulated, on a side HIP stream that waits on an event recorded on
the compute stream -- so the exchange overlaps the rest of backward.  Buckets are contiguous slices of
the optimizer's flat gradient buffer (training/optim.py) filled in reverse parameter order, the order
backward produces them.  The mean (divide by world size) is folded into the optimizer's scale pass.
BatchNorm buffers stay rank-local (the reference's per-forward buffer broadcast changes nothing for rank 0).
Works with any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU tensors in tests.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params: List[torch.nn.Parameter], offsets: List[int], flat_grad: torch.Tensor,
                 bucket_bytes: int = 25 * 1024 * 1024, process_group=None, broadcast_from: Optional[torch.Tensor] = None):
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        self.world = dist.get_world_size(process_group) if self.enabled else 1
        self.pg = process_group
        self.flat = flat_grad
        self.buckets = []  # (start, end, n_params)
        self._bucket_of = {}
        self._pending, self._handles = [], []
        self.paused = False  # True while a hipGraph owns backward: hooks stay silent, reduce_all() runs after the replay
        self.side = torch.cuda.Stream() if flat_grad.is_cuda else None
        if not self.enabled:
            return
        if broadcast_from is not None:  # DDP constructor semantics: rank 0's parameters everywhere
            dist.broadcast(broadcast_from, src=0, group=process_group)
        order = sorted(range(len(params)), key=lambda i: offsets[i], reverse=True)
        cur_hi, cur_lo, count = None, None, 0
        members = []
        for i in order:
            lo, hi = offsets[i], offsets[i] + params[i].numel()
            if cur_hi is None:
                cur_hi = hi
            cur_lo = lo
            members.append(i)
            count += 1
            if (cur_hi - cur_lo) * 4 >= bucket_bytes:
                self._close(cur_lo, cur_hi, members)
                cur_hi, members, count = None, [], 0
        if members:
            self._close(cur_lo, cur_hi, members)
        for i, p in enumerate(params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))
        self.reset()

    def _close(self, lo, hi, members):
        b = len(self.buckets)
        self.buckets.append((lo, hi, len(members)))
        for i in members:
            self._bucket_of[i] = b

    def reset(self):
        self._pending = [n for (_, _, n) in self.buckets]
        self._handles = []
        self._streams = [set() for _ in self.buckets]  # streams whose backward nodes wrote into each bucket

    def _make_hook(self, idx):
        def hook(_param):
            if self.paused:
                return
            b = self._bucket_of[idx]
            if self.flat.is_cuda:
                self._streams[b].add(torch.cuda.current_stream())
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        sl = self.flat[lo:hi]
        if self.flat.is_cuda:
            # autograd runs every node on its forward stream (AutoMoE puts the policy backbone on a side stream): the
            # launching stream first waits for every stream that accumulated into this bucket
            cur = torch.cuda.current_stream()
            for st in self._streams[b]:
                if st != cur:
                    cur.wait_stream(st)
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                self._handles.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self._handles.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def reduce_all(self):
        """One all-reduce over the whole flat gradient buffer (used after a graph replay, where per-bucket hooks did not run)."""
        if not self.enabled:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.pg)

    def finish(self):
        """Block the compute stream until every bucket is reduced; buckets whose hooks never fired (unused
        parameters) are reduced here so all ranks stay in step."""
        if not self.enabled:
            return
        for b, left in enumerate(self._pending):
            if left > 0:
                self._launch(b)
        for h in self._handles:
            h.wait()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self.reset()


class DataParallel(torch.nn.Module):
    """Thin stand-in for torch DDP's wrapper role: `.module`, forward passthrough, state_dict with the
    reference's 'module.' prefix (training/train_gating_network.py:170 saves the wrapper's state_dict)."""

    def __init__(self, module: torch.nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)
