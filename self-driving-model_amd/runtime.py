"""Process-wide numeric configuration of the HIP path.

compute dtype: torch.float16 (default; MFMA f16 with fp32 accumulation, fp32 master weights) or
torch.float32 (exact-fp32 MFMA; the parity mode that is compared with the reference's CPU fp32
arithmetic at rtol 1e-3 / atol 1e-5).
loss scale: gradients that flow through fp16 activations are multiplied by this constant where they
enter the fp16 region and divided out where they leave it (weight / BN / bias gradients), so the
fp32 parameter gradients are unscaled.  1.0 in fp32 mode.
"""
from __future__ import annotations

import contextlib
import os

# A train step keeps up to six HIP streams busy (main, policy backbone, expert graph + two expert forks, collectives); the
# runtime maps streams onto 4 hardware queues by default and streams that share one serialise.  Read when HIP initialises,
# i.e. at the first device call after this import; a value the user exported wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

_STATE = {"dtype": torch.float16, "loss_scale": 16384.0, "weight_epoch": 0, "stats_epoch": 0}


def compute_dtype() -> torch.dtype:
    return _STATE["dtype"]


def set_compute_dtype(dtype: torch.dtype, loss_scale: float | None = None):
    if dtype not in (torch.float16, torch.float32):
        raise ValueError("compute dtype must be torch.float16 or torch.float32")
    _STATE["dtype"] = dtype
    if loss_scale is not None:
        _STATE["loss_scale"] = float(loss_scale)


_INPUT_NORM = None


def set_input_normalization(mean=None, std=None) -> None:
    """Per-channel (mean, std) applied to uint8 input frames after the /255 scaling, inside the boundary layout kernel
    (train_bdd100k_ddp.py:471-473 `--imagenet_norm`: mean [0.485, 0.456, 0.406], std [0.229, 0.224, 0.225]).  None: /255 only.
    fp32 images are taken as already preprocessed, as the reference's models do."""
    global _INPUT_NORM
    _INPUT_NORM = None if mean is None else (tuple(float(v) for v in mean), tuple(float(v) for v in std))


def input_normalization():
    return _INPUT_NORM


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

_DIRECT_GRADS = False


def set_direct_grads(on: bool) -> None:
    """When on, the backward kernels of conv / Linear / LayerNorm / BatchNorm parameters accumulate straight into ``param.grad``
    (the flat fp32 gradient buffer of FusedAdamW, zeroed once per step) and autograd sees no gradient for them: saves
    a zero-fill and an add launch per parameter.  The kernels' call sites report finished gradients through grad_ready()."""
    global _DIRECT_GRADS
    _DIRECT_GRADS = bool(on)


def direct_grads() -> bool:
    return _DIRECT_GRADS


# Listeners told that a parameter's gradient is complete for this backward (training/ddp.py GradBucketReducer).  autograd's
# post-accumulate hooks cover parameters whose gradient autograd accumulates; a backward kernel that added straight into
# ``param.grad`` (direct mode) reports it here instead, so the bucketed all-reduce sees every parameter exactly once.
_GRAD_LISTENERS = []


def add_grad_listener(cb) -> None:
    import weakref
    _GRAD_LISTENERS.append(weakref.WeakMethod(cb) if hasattr(cb, "__self__") else (lambda: cb))


def grad_ready(*params) -> None:
    if not _GRAD_LISTENERS:
        return
    alive = []
    for ref in _GRAD_LISTENERS:
        cb = ref()
        if cb is None:
            continue
        alive.append(ref)
        for p in params:
            if p is not None:
                cb(p)
    _GRAD_LISTENERS[:] = alive


def loss_scale() -> float:
    return _STATE["loss_scale"] if _STATE["dtype"] == torch.float16 else 1.0


def set_loss_scale(v: float):
    _STATE["loss_scale"] = float(v)


def weight_epoch() -> int:
    """Bumped by the fused optimizer after every in-place parameter update made through raw pointers
    (which torch's tensor version counter does not see); packed-weight caches key on it."""
    return _STATE["weight_epoch"]


def bump_weight_epoch():
    _STATE["weight_epoch"] += 1


def stats_epoch() -> int:
    """Bumped whenever BatchNorm running statistics may have been updated through raw pointers: by every train-mode
    am_bn_finalize issued from Python and by every replay of a captured train step (a replay runs no Python).  Caches of
    anything derived from the running statistics (the eval-mode conv+BN fold, hip/conv.py) key on it."""
    return _STATE["stats_epoch"]


def bump_stats_epoch():
    _STATE["stats_epoch"] += 1


@contextlib.contextmanager
def precision(dtype: torch.dtype, loss_scale: float | None = None):
    old = dict(_STATE)
    set_compute_dtype(dtype, loss_scale)
    try:
        yield
    finally:
        _STATE.update({"dtype": old["dtype"], "loss_scale": old["loss_scale"]})


# ---- per-step zeroed fp64 arena for BatchNorm statistics ---------------------------------------------------------
# Every conv+BN layer needs a small zeroed fp64 accumulator (forward sums, backward sums).  Allocating and zero-filling
# ~130 of them per step costs more launches than the arithmetic; instead one buffer is zeroed by a single memset at
# begin_step() and handed out in slices.  A slice is only valid until the next begin_step().
# Two arenas: "main" for a whole model step, "experts" for the frozen-expert phase when a trainer runs it as its own graph
# one step ahead of the rest (training/train_gating_network.py): the two phases of neighbouring steps then overlap on the
# GPU and must not share accumulators.
_ARENAS = {"main": {"buf": None, "off": 0, "cap": 1 << 22}, "experts": {"buf": None, "off": 0, "cap": 1 << 22}}
_PHASE = {"name": "main"}


_STEP = {"counter": None}


def step_counter(device) -> torch.Tensor:
    """Device int64 incremented by begin_step(): kernels that need per-step randomness (dropout) read it on the device,
    which keeps them correct under hipGraph replay where host-side counters are frozen."""
    if _STEP["counter"] is None or _STEP["counter"].device != torch.device(device):
        _STEP["counter"] = torch.zeros(1, dtype=torch.int64, device=device)
    return _STEP["counter"]


def begin_step(device=None, phase: str = "main"):
    """Zero the phase's statistics arena (one memset), make it the current one and (phase "main") bump the device step
    counter.  Called at the start of every model forward / train step."""
    a = _ARENAS[phase]
    _PHASE["name"] = phase
    if device is not None and phase == "main":
        step_counter(device).add_(1)
    if a["buf"] is None or (device is not None and a["buf"].device != torch.device(device)):
        a["buf"] = torch.zeros(a["cap"], dtype=torch.float64, device=device if device is not None else "cuda")
    else:
        a["buf"].zero_()
    a["off"] = 0


def arena_zeros(n: int, device) -> torch.Tensor:
    a = _ARENAS[_PHASE["name"]]
    if a["buf"] is None or a["buf"].device != torch.device(device) or a["off"] + n > a["cap"]:
        return torch.zeros(n, dtype=torch.float64, device=device)  # outside a step (or exhausted): plain allocation
    out = a["buf"][a["off"]:a["off"] + n]
    a["off"] += (n + 1) & ~1
    return out
