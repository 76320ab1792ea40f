"""Data-parallel gradient exchange for one process per GPU (the reference wraps its model in
torch DistributedDataParallel: training/train_bdd100k_ddp.py:497, training/train_gating_network.py:236).

One RCCL all-reduce (sum) per gradient bucket over xGMI, issued as soon as every gradient in the bucket is
complete, on a side HIP stream that waits on an event recorded on the compute stream -- so the exchange
overlaps the rest of backward.  Buckets are contiguous slices of the optimizer's flat gradient buffer
(training/optim.py) filled in reverse parameter order, the order backward produces them.  The mean (divide by
world size) is folded into the optimizer's scale pass.  BatchNorm buffers stay rank-local (the reference's
per-forward buffer broadcast changes nothing for rank 0).

"Gradient complete" comes from two sources: autograd's post-accumulate hook (gradients autograd accumulates: conv
weights / biases) and runtime.grad_ready() from the backward kernels that add straight into ``param.grad``
(Linear / LayerNorm / BatchNorm parameters in direct mode, hip/ops.py, hip/conv.py).  A parameter counts once per
backward (its first report); parameters are not shared between layers on this path.

hipGraph steps: with the "nccl" backend (= RCCL on ROCm) the bucket collectives are CAPTURED with the step --
event record on the compute stream, wait + all-reduce on the side stream, join before the optimizer -- so a
replay exchanges each bucket beside the rest of backward exactly as the eager step does.  Backends that cannot
be captured ("gloo" in the CPU / shared-GPU tests) keep the hooks silent during the capture (`paused`) and run
ONE all-reduce over the flat buffer after the replay (`reduce_all`).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import runtime


class GradBucketReducer:
    def __init__(self, params: List[torch.nn.Parameter], offsets: List[int], flat_grad: torch.Tensor,
                 bucket_bytes: Optional[int] = None, process_group=None, broadcast_from: Optional[torch.Tensor] = None,
                 force: Optional[bool] = None):
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("AUTOMOE_BUCKET_MB", "25")) * 1024 * 1024)
        if force is None:  # tests: run the bucket machinery on a 1-rank group (one GPU box, RCCL itself still executes)
            force = os.environ.get("AUTOMOE_DDP_FORCE", "0") == "1"
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if live else 1
        self.enabled = live and (self.world > 1 or force)
        self.pg = process_group
        self.flat = flat_grad
        self.buckets = []  # (start, end, n_params)
        self._bucket_of = {}
        self._index = {}
        self._pending, self._handles = [], []
        self.paused = False  # True while a hipGraph without captured collectives owns backward: reduce_all() runs after the replay
        self.side = torch.cuda.Stream() if flat_grad.is_cuda else None
        # collectives of this backend can be recorded into a hipGraph (c10d's NCCL work is stream-ordered; gloo's is host-side)
        self.capturable = bool(self.enabled and flat_grad.is_cuda and dist.get_backend(process_group) == "nccl"
                               and os.environ.get("AUTOMOE_GRAPH_ALLREDUCE", "1") != "0")
        if not self.enabled:
            return
        if broadcast_from is not None:  # DDP constructor semantics: rank 0's parameters everywhere
            dist.broadcast(broadcast_from, src=0, group=process_group)
        order = sorted(range(len(params)), key=lambda i: offsets[i], reverse=True)
        cur_hi, cur_lo = None, None
        members = []
        for i in order:
            lo, hi = offsets[i], offsets[i] + params[i].numel()
            if cur_hi is None:
                cur_hi = hi
            cur_lo = lo
            members.append(i)
            if (cur_hi - cur_lo) * 4 >= bucket_bytes:
                self._close(cur_lo, cur_hi, members)
                cur_hi, members = None, []
        if members:
            self._close(cur_lo, cur_hi, members)
        for i, p in enumerate(params):
            self._index[id(p)] = i
            p.register_post_accumulate_grad_hook(self._on_grad)
        self._params = params  # keeps id() keys valid
        runtime.add_grad_listener(self._on_grad)
        self.reset()

    def _close(self, lo, hi, members):
        b = len(self.buckets)
        self.buckets.append((lo, hi, len(members)))
        for i in members:
            self._bucket_of[i] = b

    def reset(self):
        self._pending = [n for (_, _, n) in self.buckets]
        self._handles = []
        self._seen = bytearray(len(self._index))
        self._streams = [set() for _ in self.buckets]  # streams whose backward nodes wrote into each bucket

    def _on_grad(self, param):
        """`param`'s gradient for this backward is complete in the flat buffer (autograd hook or runtime.grad_ready)."""
        if self.paused or not self.enabled:
            return
        idx = self._index.get(id(param))
        if idx is None:
            return
        b = self._bucket_of[idx]
        if self.flat.is_cuda:
            self._streams[b].add(torch.cuda.current_stream())
        # a parameter written in direct mode is announced by its kernel's call site AND (torch 2.10 runs the post-accumulate
        # hook even for a gradient the node returned as None) by autograd; both come after its only contribution of this
        # backward was enqueued, so the first report counts and the second is dropped
        if self._seen[idx]:
            return
        self._seen[idx] = 1
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

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
        """One all-reduce over the whole flat gradient buffer (after a graph replay whose capture held no collectives)."""
        if not self.enabled:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.pg)

    def finish(self):
        """Block the compute stream until every bucket is reduced; buckets that never completed (unused
        parameters) are reduced here so all ranks stay in step.  Inside a capture this records the joins."""
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

    # ---- protocol shared by the trainers' hipGraph steps ---------------------------------------------------------
    def capture_begin(self) -> bool:
        """Call before capturing a step.  True: the capture records the bucket collectives (call finish() inside it, after
        backward); False: hooks are silenced for the capture and every step ends with reduce_all().
        A gradient buffer that is ONE bucket (the 11.5 MB of the frozen-expert gating stage) has nothing to overlap -- its only
        collective starts when backward ends either way -- so it takes the plain post-replay all-reduce unless
        AUTOMOE_GRAPH_ALLREDUCE=1 asks for the captured form; several buckets (trainable experts: 49-160 MB) are captured."""
        in_graph = self.enabled and self.capturable and (len(self.buckets) > 1 or os.environ.get("AUTOMOE_GRAPH_ALLREDUCE") == "1")
        self.paused = self.enabled and not in_graph
        self.reset()
        return in_graph

    def agree(self, mode: int) -> int:
        """Ranks must issue matching collectives: MIN over ranks of the step mode each one reached
        (2 graph with captured collectives, 1 graph + reduce_all, 0 eager)."""
        if not self.enabled:
            return mode
        t = torch.tensor([mode], device=self.flat.device, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.pg)
        return int(t.item())


def capture_error_mode() -> str:
    """`capture_error_mode` of every hipGraph capture in this package: "thread_local", whether or not this trainer's own reducer
    is enabled (round 2 chose "global" whenever the reducer was off, although another object's process group -- and c10d's
    watchdog thread, which polls events with hipEventQuery about every 100 ms -- may be alive in the same process).

    What is known about round 2's one abort ("a hipEventQuery inside another thread's capture window came back with a HIP error"):
    it was seen once, no log of it was kept, and it does NOT reproduce: scratch/capture_mode_probe.py (a second thread calling
    Event.query() every 2 ms through a capture window, 40 queries, one child process per mode) ran clean in thread_local, relaxed
    AND global mode on this runtime (gpurun_out/r3_capture_probe.log).  So a foreign-thread query is legal
    here in every mode and the 0.3 s sleep that used to sit in quiesce_collectives guarded against something that was never
    established: it is removed.  thread_local is kept because it is the mode whose CONTRACT allows other threads' runtime calls
    (CUDA programming guide, stream-capture modes), not because a failure was tied to the other two.  If a worker aborts again,
    tests/test_hip_multigpu.py now keeps its whole output, c10d's flight-recorder dump and C++ stacks under gpurun_out/."""
    return "thread_local"


def quiesce_collectives(reducer: "GradBucketReducer"):
    """Before a capture begins: every eager collective this reducer issued (earlier steps' buckets) has been waited for and the
    device is idle -- ordering hygiene that is true by construction (handles held and waited, then a device synchronize), no
    timing margin."""
    if reducer.enabled and torch.cuda.is_available():
        for h in reducer._handles:
            h.wait()
        torch.cuda.synchronize()


def capture_step(reducer: GradBucketReducer, capture_fn, what: str = "train step"):
    """Capture one training step into a hipGraph in the best mode every rank can reach.

    capture_fn(in_graph) -> (graph, result) captures zero_grad + forward + loss + backward and, when `in_graph`, the
    reducer's finish() (bucket collectives recorded with the step).  Returns (graph, result, in_graph); graph is None when
    the step stays eager.  Ranks agree after every attempt (a rank whose capture failed pulls all of them one mode down),
    so every rank issues the same collectives afterwards."""
    import traceback
    import warnings
    want = 2 if reducer.capture_begin() else 1
    graph = result = None
    while want > 0:
        reducer.paused = reducer.enabled and want == 1
        reducer.reset()
        got = want
        quiesce_collectives(reducer)
        try:
            graph, result = capture_fn(want == 2)
        except Exception as e:  # noqa: BLE001  (capture is an optimisation: fall back, loudly)
            warnings.warn(f"hipGraph capture of the {what} failed in mode {want} ({e!r})\n"
                          + "".join(traceback.format_exc().splitlines(True)[-14:]))
            graph = result = None
            got = want - 1
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        agreed = reducer.agree(got)
        if agreed == want:
            break
        want, graph, result = agreed, None, None
    in_graph = graph is not None and want == 2
    reducer.paused = reducer.enabled and graph is not None and not in_graph
    reducer.reset()
    return graph, result, in_graph


class StepStream:
    """Every step of a trainer -- the eager ones before the capture, the capture itself, replays and eager fallbacks -- runs on ONE
    side HIP stream owned by the trainer.

    Why: autograd binds a parameter's AccumulateGrad node to the stream that was current when the node was created, and the node
    lives as long as any tensor of that iteration's graph does (a loss a caller keeps).  If such a node from an eager step on the
    DEFAULT stream is still alive when the step is captured on the capture stream, the captured backward accumulates on the
    default stream behind an event of the capturing stream: the default stream is pulled into the capture and never joined.
    CUDA reports cudaErrorStreamCaptureUnjoined; HIP (ROCm 7.2) takes a host segfault inside hipStreamEndCapture
    (scratch/repro_capture_segv.py: "stale" faults, "clean" and "samestream" capture fine).  With all steps on the same
    stream a surviving node is bound to the capture stream itself.  Trainers also hand out DETACHED loss tensors, so callers
    cannot keep a graph alive by keeping a loss."""

    def __init__(self, device):
        dev = torch.device(device)
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    def __enter__(self):
        if self.stream is None:
            return self
        self._outer = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_stream(self._outer)  # inputs the caller produced
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.stream is None:
            return False
        self._ctx.__exit__(*exc)
        self._outer.wait_stream(self.stream)  # results are ordered for the caller's stream; nothing blocks the host
        return False


def detached(losses):
    """The same values without the autograd graph (see StepStream)."""
    if isinstance(losses, dict):
        return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in losses.items()}
    return losses.detach() if isinstance(losses, torch.Tensor) else losses


class DataParallel(torch.nn.Module):
    """Thin stand-in for torch DDP's wrapper role: `.module`, forward passthrough, state_dict with the
    reference's 'module.' prefix (training/train_gating_network.py:170 saves the wrapper's state_dict)."""

    def __init__(self, module: torch.nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *a, **kw):
        return self.module(*a, **kw)
