"""autograd nodes over the C ABI for everything on the path that is not the conv itself:
layout changes at the NCHW-fp32 boundary, max-pool, global average pools, bilinear upsample,
pixel-wise cross entropy, and the fp32 MoE-tail ops (linear, LayerNorm, dropout, gate combine).

Every node calls libautomoe_hip.so on PyTorch's current HIP stream; none has a torch fallback.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import lib as _lib
from .conv import dt_code, ptr, require_hip, stream


def _L():
    return _lib.get()


def _runtime():
    from .. import runtime
    return runtime


def _grad_ready(p) -> bool:
    """Parameter with a preallocated, dense fp32 ``.grad`` on the same device (FusedAdamW's flat buffer views)."""
    g = getattr(p, "grad", None)
    return g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == p.device


# --------------------------------------------------------------------------------------------
# boundary layout
# --------------------------------------------------------------------------------------------
def image_to_s2d(img: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[B,3,H,W] fp32 NCHW (the reference's batch['image']) -- or uint8 raw frames, which are scaled by 1/255 and, when
    runtime.set_input_normalization(mean, std) is set, normalised exactly like the reference's loader (SURVEY.md section 8(f)
    row 4) -- -> space-to-depth(2) NHWC [B,ceil(H/2),ceil(W/2),16] `dtype`
    (channel (py*2+px)*3 + c = img[b,c,2Y+py,2X+px]); the 3-channel stride-2 first layers run on it as stride-1 convs with
    16 input channels.  The original size rides along as `.orig_hw`.  No gradient: the image is never a leaf that requires
    grad on this path.  Odd sizes are zero-padded by one row/column, which the convs' own zero padding makes exact."""
    require_hip(img, "image")
    img = img.detach()
    raw_u8 = img.dtype == torch.uint8
    if not raw_u8 and img.dtype != torch.float32:
        img = img.float()
    B, C, H, W = img.shape
    if C > 4:
        raise ValueError(f"image has {C} channels; the first-layer path handles at most 4")
    if ((H & 1) or (W & 1)) and not raw_u8:
        img = torch.nn.functional.pad(img, (0, W & 1, 0, H & 1))
    img = img.contiguous()
    H2, W2 = (img.shape[2] + 1) // 2, (img.shape[3] + 1) // 2
    out = torch.empty((B, H2, W2, 16), dtype=dtype, device=img.device)
    if raw_u8:
        # raw camera frames: /255 and the optional ImageNet normalisation happen inside the layout kernel
        norm = _runtime().input_normalization()
        mean = (ctypes.c_float * C)(*norm[0][:C]) if norm is not None else None
        std = (ctypes.c_float * C)(*norm[1][:C]) if norm is not None else None
        _L().am_image_u8_s2d(dt_code(dtype), ptr(img), ptr(out), B, C, img.shape[2], img.shape[3], mean, std, stream())
    else:
        _L().am_image_s2d(dt_code(dtype), ptr(img), ptr(out), B, C, img.shape[2], img.shape[3], stream())
    out.orig_hw = (H, W)
    return out


image_to_nhwc = image_to_s2d  # the boundary layout every backbone entry point expects


class NhwcToNchw(torch.autograd.Function):
    """[B,h,w,ld] NHWC `dtype` -> [B,C,h,w] fp32 NCHW (expert head outputs handed to the trainer).
    Backward converts the fp32 gradient back and applies the loss scale of the fp16 region."""

    @staticmethod
    def forward(ctx, x, C: int, loss_scale: float):
        B, H, W, ld = x.shape
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        _L().am_nhwc_to_nchw(dt_code(x.dtype), ptr(x), ptr(out), B, C, H, W, ld, 1.0, stream())
        ctx.meta = (x.dtype, ld, loss_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        dtype, ld, ls = ctx.meta
        g = g.contiguous()
        B, C, H, W = g.shape
        dx = torch.empty((B, H, W, ld), dtype=dtype, device=g.device)
        _L().am_nchw_to_nhwc(dt_code(dtype), ptr(g), ptr(dx), B, C, H, W, ld, float(ls), stream())
        return dx, None, None


class MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, IH, IW, C = x.shape
        OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
        y = torch.empty((B, OH, OW, C), dtype=x.dtype, device=x.device)
        need = x.requires_grad
        arg = torch.empty((B, OH, OW, C), dtype=torch.uint8, device=x.device) if need else None
        _L().am_maxpool3x3s2_fwd(dt_code(x.dtype), ptr(x), ptr(y), ptr(arg), B, IH, IW, C, stream())
        ctx.shape = (B, IH, IW, C)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, IH, IW, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((B, IH, IW, C), dtype=dy.dtype, device=dy.device)
        _L().am_maxpool3x3s2_bwd(dt_code(dy.dtype), ptr(dy), ptr(arg), ptr(dx), B, IH, IW, C, stream())
        return dx


class GapNhwc(torch.autograd.Function):
    """AdaptiveAvgPool2d(1)+flatten on NHWC `dtype` -> [B,C] fp32."""

    @staticmethod
    def forward(ctx, x, loss_scale: float):
        B, H, W, C = x.shape
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        _L().am_gap_nhwc_fwd(dt_code(x.dtype), ptr(x), C, ptr(out), B, H * W, C, stream())
        ctx.meta = (x.dtype, B, H, W, C, loss_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        dtype, B, H, W, C, ls = ctx.meta
        g = g.contiguous()
        dx = torch.empty((B, H, W, C), dtype=dtype, device=g.device)
        _L().am_gap_nhwc_bwd(dt_code(dtype), ptr(g), ptr(dx), C, B, H * W, C, float(ls), stream())
        return dx, None


class GapPlane(torch.autograd.Function):
    """AdaptiveAvgPool2d(1)+flatten on NCHW fp32 -> [B,C] (the extractors' first op)."""

    @staticmethod
    def forward(ctx, x):
        require_hip(x, "extractor input")
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        _L().am_gap_plane_fwd(ptr(x), ptr(out), B * C, H * W, stream())
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        g = g.contiguous()
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
        _L().am_gap_plane_bwd(ptr(g), ptr(dx), B * C, H * W, stream())
        return dx


class BilinearUp(torch.autograd.Function):
    """F.interpolate(bilinear, align_corners=False): [B,h,w,ld] NHWC `dtype` -> [B,C,H,W] fp32."""

    @staticmethod
    def forward(ctx, low, C: int, H: int, W: int, loss_scale: float):
        B, h, w, ld = low.shape
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=low.device)
        _L().am_bilinear_up_fwd(dt_code(low.dtype), ptr(low), ld, ptr(out), B, C, h, w, H, W, stream())
        ctx.meta = (low.dtype, B, C, h, w, ld, H, W, loss_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        dtype, B, C, h, w, ld, H, W, ls = ctx.meta
        g = g.contiguous()
        dlow = (torch.zeros if ld != C else torch.empty)((B, h, w, ld), dtype=dtype, device=g.device)
        _L().am_bilinear_up_bwd(dt_code(dtype), ptr(g), ptr(dlow), ld, B, C, h, w, H, W, float(ls), None, stream())
        return dlow, None, None, None, None


_COLSUM_CACHE = {}


def _colsum(in_size: int, out_size: int, device) -> torch.Tensor:
    key = (in_size, out_size, str(device))
    if key not in _COLSUM_CACHE:
        c = torch.empty(in_size, dtype=torch.float32, device=device)
        _L().am_bilinear_colsum(ptr(c), in_size, out_size, stream())
        _COLSUM_CACHE[key] = c
    return _COLSUM_CACHE[key]


class UpsampleGap(torch.autograd.Function):
    """AdaptiveAvgPool2d(1)(F.interpolate(low, (H,W), bilinear)) -> [B,C] fp32 straight from the low-resolution NHWC
    logits: the mean of a bilinear upsample is a fixed separable weighted sum of the low-resolution map."""

    @staticmethod
    def forward(ctx, low, C: int, H: int, W: int, loss_scale: float):
        B, h, w, ld = low.shape
        cy, cx = _colsum(h, H, low.device), _colsum(w, W, low.device)
        out = torch.empty((B, C), dtype=torch.float32, device=low.device)
        _L().am_upsample_gap_fwd(dt_code(low.dtype), ptr(low), ld, ptr(cy), ptr(cx), ptr(out), B, C, h, w, stream())
        ctx.meta = (low.dtype, B, C, h, w, ld, loss_scale)
        ctx.save_for_backward(cy, cx)
        return out

    @staticmethod
    def backward(ctx, g):
        cy, cx = ctx.saved_tensors
        dtype, B, C, h, w, ld, ls = ctx.meta
        g = g.contiguous()
        dlow = torch.empty((B, h, w, ld), dtype=dtype, device=g.device)
        _L().am_upsample_gap_bwd(dt_code(dtype), ptr(g), ptr(cy), ptr(cx), ptr(dlow), ld, B, C, h, w, float(ls), stream())
        return dlow, None, None, None, None


class CrossEntropy2d(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index) on [B,C,H,W] fp32 logits / [B,H,W] int64 targets -> scalar (mean over
    valid pixels).  The valid count never leaves the device."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index: int):
        require_hip(logits, "logits")
        logits = logits.contiguous()
        target = target.contiguous()
        if target.dtype != torch.int64:
            target = target.long()
        B, C = logits.shape[0], logits.shape[1]
        HW = logits[0, 0].numel()
        acc = torch.empty(2, dtype=torch.float64, device=logits.device)
        _L().am_ce2d_fwd(ptr(logits), ptr(target), B, C, HW, int(ignore_index), ptr(acc), stream())
        ctx.save_for_backward(logits, target, acc)
        ctx.meta = (B, C, HW, int(ignore_index))
        return (acc[0] / acc[1]).float()  # 0/0 -> nan when every pixel is ignored, as torch

    @staticmethod
    def backward(ctx, g):
        logits, target, acc = ctx.saved_tensors
        B, C, HW, ign = ctx.meta
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _L().am_ce2d_bwd(ptr(logits), ptr(target), B, C, HW, ign, ptr(acc), ptr(g), ptr(d), stream())
        return d, None, None


class UpsampleCrossEntropy(torch.autograd.Function):
    """CrossEntropy2d(BilinearUp(low), target) without the [B,C,H,W] logits (am_upsample_ce2d_*): the dense experts' training
    loss straight from the low-resolution NHWC logits.  ``supported(C)`` tells whether the class count has a kernel."""

    @staticmethod
    def supported(C: int) -> bool:
        return C in (3, 19)

    @staticmethod
    def forward(ctx, low, target, C: int, H: int, W: int, ignore_index: int, loss_scale: float):
        require_hip(low, "low-resolution logits")
        low = low.contiguous()
        target = target.contiguous()
        if target.dtype != torch.int64:
            target = target.long()
        B, h, w, ld = low.shape
        if tuple(target.shape) != (B, H, W):
            raise ValueError(f"target shape {tuple(target.shape)} does not match logits upsampled to {(B, H, W)}")
        acc = torch.empty(2, dtype=torch.float64, device=low.device)
        G = torch.empty((B, h, w, C), dtype=torch.float32, device=low.device)
        _L().am_upsample_ce2d_fwd(dt_code(low.dtype), ptr(low), ld, ptr(target), B, C, h, w, H, W, int(ignore_index), ptr(acc), ptr(G), stream())
        ctx.save_for_backward(G, acc)
        ctx.meta = (low.dtype, B, C, h, w, ld, loss_scale)
        return (acc[0] / acc[1]).float()  # 0/0 -> nan when every pixel is ignored, as torch

    @staticmethod
    def backward(ctx, g):
        G, acc = ctx.saved_tensors
        dtype, B, C, h, w, ld, ls = ctx.meta
        g = g.contiguous().float()
        dlow = (torch.zeros if ld != C else torch.empty)((B, h, w, ld), dtype=dtype, device=g.device)
        _L().am_upsample_ce2d_bwd(dt_code(dtype), ptr(G), ptr(acc), ptr(g), float(ls), ptr(dlow), ld, B, C, h, w, stream())
        return dlow, None, None, None, None, None, None


# --------------------------------------------------------------------------------------------
# fp32 MoE tail
# --------------------------------------------------------------------------------------------
class LinearAct(torch.autograd.Function):
    """y = act(x W^T + b) for nn.Linear weights; optional fused ReLU."""

    @staticmethod
    def forward(ctx, x, W, b, relu: bool):
        require_hip(x, "linear input")
        x = x.contiguous()
        M, K = x.shape
        N = W.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _L().am_linear_fwd(ptr(x), K, ptr(W), ptr(b), ptr(y), N, M, N, K, int(relu), stream())
        ctx.relu = relu
        ctx.save_for_backward(x, W, y if relu else None)
        ctx.has_b = b is not None
        ctx.params = (W, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = W.shape[0]
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _L().am_linear_bwd_input(ptr(dy), N, ptr(y), N, ptr(W), ptr(dx), K, M, N, K, 0, stream())
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            Wp, bp = ctx.params
            if (_runtime().direct_grads() and ctx.needs_input_grad[1] and _grad_ready(Wp)
                    and (not ctx.has_b or (ctx.needs_input_grad[2] and _grad_ready(bp)))):
                # the kernel accumulates (+=): straight into the optimizer's gradient buffer, nothing for autograd to add
                _L().am_linear_bwd_weight(ptr(dy), N, ptr(y), N, ptr(x), K, ptr(Wp.grad), ptr(bp.grad) if ctx.has_b else None,
                                          M, N, K, stream())
                _runtime().grad_ready(Wp, bp if ctx.has_b else None)
                return dx, None, None, None
            dW = torch.zeros_like(W)
            db = torch.zeros(N, dtype=torch.float32, device=W.device) if ctx.has_b else None
            _L().am_linear_bwd_weight(ptr(dy), N, ptr(y), N, ptr(x), K, ptr(dW), ptr(db), M, N, K, stream())
        return dx, dW, db, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        require_hip(x, "layernorm input")
        x = x.contiguous()
        M, D = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _L().am_layernorm_fwd(ptr(x), D, ptr(gamma), ptr(beta), float(eps), ptr(y), D, ptr(mean), ptr(rstd), M, D, stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        M, D = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gp, bp = ctx.params
        if _runtime().direct_grads() and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and _grad_ready(gp) and _grad_ready(bp):
            _L().am_layernorm_bwd(ptr(dy), D, ptr(x), D, ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), D, ptr(gp.grad), ptr(bp.grad), M, D, stream())
            _runtime().grad_ready(gp, bp)
            return dx, None, None, None
        dg = torch.zeros_like(gamma) if need_p else None
        db = torch.zeros_like(gamma) if need_p else None
        _L().am_layernorm_bwd(ptr(dy), D, ptr(x), D, ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), D, ptr(dg), ptr(db), M, D, stream())
        return dx, dg, db, None


class GroupedLinear(torch.autograd.Function):
    """G independent layers y_i = dropout_p(relu?(x_i W_i^T + b_i)) in ONE launch (am_moe_tail_linear_fwd); backward in two
    (input gradients of all members, parameter gradients of all members).  The reference's Linear -> ReLU -> Dropout triples of
    the MoE tail (expert_extractors.py:30-34, context_features.py:143-149, gating_network.py:13-20,38-44, trajectory_head.py:44-53)
    run as the epilogue of the Linear; the dropout mask is never stored (y == 0 <=> dropped or rectified).
    apply(spec, x_0, W_0, b_0, x_1, W_1, b_1, ...) -> (y_0, y_1, ...); spec = [(relu, drop_p), ...] (drop_p 0 in eval mode)."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        global _DROPOUT_CALLS
        from . import lib as _lib
        G = len(spec)
        assert len(tensors) == 3 * G and 1 <= G <= _lib.AM_TAIL_MAX_GROUP
        xs = [tensors[3 * i].contiguous() for i in range(G)]
        Ws, bs = [tensors[3 * i + 1] for i in range(G)], [tensors[3 * i + 2] for i in range(G)]
        require_hip(xs[0], "linear input")
        M = xs[0].shape[0]
        ys = [torch.empty((M, W.shape[0]), dtype=torch.float32, device=xs[0].device) for W in Ws]
        arr = (_lib.TailLinear * G)()
        seeds = []
        for i, (relu, p_drop) in enumerate(spec):
            assert xs[i].shape == (M, Ws[i].shape[1]) and xs[i].dtype == torch.float32
            d = arr[i]
            d.x, d.W, d.bias, d.y = ptr(xs[i]), ptr(Ws[i]), ptr(bs[i]), ptr(ys[i])
            d.ldx, d.ldy, d.N, d.K, d.relu = xs[i].shape[1], ys[i].shape[1], Ws[i].shape[0], Ws[i].shape[1], int(relu)
            d.drop_p = float(p_drop)
            if p_drop > 0.0:
                _DROPOUT_CALLS += 1
                d.seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _DROPOUT_CALLS * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
            seeds.append(int(d.seed))
        step = _runtime().step_counter(xs[0].device) if any(p_ > 0.0 for _, p_ in spec) else None
        _L().am_moe_tail_linear_fwd(arr, G, M, ptr(step), stream())
        ctx.spec, ctx.G, ctx.M = list(spec), G, M
        ctx.params = [(Ws[i], bs[i]) for i in range(G)]
        ctx.save_for_backward(*xs, *Ws, *[ys[i] if (spec[i][0] or spec[i][1] > 0.0) else xs[i].new_empty(0) for i in range(G)])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        from . import lib as _lib
        G, M = ctx.G, ctx.M
        saved = ctx.saved_tensors
        xs, Ws, yacts = saved[:G], saved[G:2 * G], saved[2 * G:]
        arr = (_lib.TailLinear * G)()
        grads = [None] * (3 * G)
        keep = []  # tensors the launches read: alive until this function returns (same stream: the allocator orders reuse)
        direct_done = []
        n_active = 0
        for i in range(G):
            relu, p_drop = ctx.spec[i]
            d = arr[i]
            N, K = Ws[i].shape
            dy = dys[i]
            if dy is None:
                dy = torch.zeros((M, N), dtype=torch.float32, device=xs[i].device)
            dy = dy.contiguous()
            keep.append(dy)
            d.x, d.W, d.dy = ptr(xs[i]), ptr(Ws[i]), ptr(dy)
            d.ldx, d.lddy, d.N, d.K = K, N, N, K
            d.gscale = 1.0 / (1.0 - p_drop) if p_drop > 0.0 else 1.0
            if yacts[i].numel():
                d.yact, d.ldya = ptr(yacts[i]), N
            if ctx.needs_input_grad[1 + 3 * i]:
                dx = torch.empty_like(xs[i])
                d.dx, d.lddx, d.dx_accumulate = ptr(dx), K, 0
                grads[3 * i] = dx
            Wp, bp = ctx.params[i]
            need_w = ctx.needs_input_grad[2 + 3 * i]
            need_b = bp is not None and ctx.needs_input_grad[3 + 3 * i]
            if need_w or need_b:
                if _runtime().direct_grads() and need_w and _grad_ready(Wp) and (bp is None or (need_b and _grad_ready(bp))):
                    d.dW, d.dbias = ptr(Wp.grad), ptr(bp.grad) if bp is not None else None
                    direct_done.append((Wp, bp))
                else:
                    dW = torch.zeros_like(Ws[i])
                    db = torch.zeros(N, dtype=torch.float32, device=Ws[i].device) if bp is not None else None
                    d.dW, d.dbias = ptr(dW), ptr(db)
                    grads[3 * i + 1], grads[3 * i + 2] = dW, db
            n_active += 1
        _L().am_moe_tail_linear_bwd(arr, G, M, stream())
        for Wp, bp in direct_done:
            _runtime().grad_ready(Wp, bp)
        return (None, *grads)


class GroupedLayerNorm(torch.autograd.Function):
    """G independent LayerNorms in one launch (am_moe_tail_layernorm_fwd); backward (input + parameter gradients) in one.
    apply(eps_list, x_0, gamma_0, beta_0, x_1, ...) -> (y_0, y_1, ...)."""

    @staticmethod
    def forward(ctx, eps_list, *tensors):
        from . import lib as _lib
        G = len(eps_list)
        assert len(tensors) == 3 * G and 1 <= G <= _lib.AM_TAIL_MAX_GROUP
        xs = [tensors[3 * i].contiguous() for i in range(G)]
        gs, bs = [tensors[3 * i + 1] for i in range(G)], [tensors[3 * i + 2] for i in range(G)]
        require_hip(xs[0], "layernorm input")
        M = xs[0].shape[0]
        ys = [torch.empty_like(x) for x in xs]
        stats = torch.empty((G, 2, M), dtype=torch.float32, device=xs[0].device)
        arr = (_lib.TailLayerNorm * G)()
        for i in range(G):
            d = arr[i]
            D = xs[i].shape[1]
            d.x, d.gamma, d.beta, d.y, d.mean, d.rstd = ptr(xs[i]), ptr(gs[i]), ptr(bs[i]), ptr(ys[i]), ptr(stats[i, 0]), ptr(stats[i, 1])
            d.ldx, d.ldy, d.D, d.eps = D, D, D, float(eps_list[i])
        _L().am_moe_tail_layernorm_fwd(arr, G, M, stream())
        ctx.G, ctx.M, ctx.eps = G, M, list(eps_list)
        ctx.params = [(gs[i], bs[i]) for i in range(G)]
        ctx.save_for_backward(stats, *xs, *gs)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        from . import lib as _lib
        G, M = ctx.G, ctx.M
        stats, *rest = ctx.saved_tensors
        xs, gs = rest[:G], rest[G:]
        arr = (_lib.TailLayerNorm * G)()
        grads = [None] * (3 * G)
        keep, direct_done = [], []
        for i in range(G):
            d = arr[i]
            D = xs[i].shape[1]
            dy = dys[i] if dys[i] is not None else torch.zeros_like(xs[i])
            dy = dy.contiguous()
            keep.append(dy)
            d.x, d.gamma, d.mean, d.rstd, d.dy = ptr(xs[i]), ptr(gs[i]), ptr(stats[i, 0]), ptr(stats[i, 1]), ptr(dy)
            d.ldx, d.lddy, d.D, d.eps = D, D, D, float(ctx.eps[i])
            if ctx.needs_input_grad[1 + 3 * i]:
                dx = torch.empty_like(xs[i])
                d.dx, d.lddx = ptr(dx), D
                grads[3 * i] = dx
            gp, bp = ctx.params[i]
            if ctx.needs_input_grad[2 + 3 * i] or ctx.needs_input_grad[3 + 3 * i]:
                if (_runtime().direct_grads() and ctx.needs_input_grad[2 + 3 * i] and ctx.needs_input_grad[3 + 3 * i]
                        and _grad_ready(gp) and _grad_ready(bp)):
                    d.dgamma, d.dbeta = ptr(gp.grad), ptr(bp.grad)
                    direct_done.append((gp, bp))
                else:
                    dg, db = torch.zeros_like(gs[i]), torch.zeros_like(gs[i])
                    d.dgamma, d.dbeta = ptr(dg), ptr(db)
                    grads[3 * i + 1], grads[3 * i + 2] = dg, db
        _L().am_moe_tail_layernorm_bwd(arr, G, M, stream())
        for gp, bp in direct_done:
            _runtime().grad_ready(gp, bp)
        return (None, *grads)


_DROPOUT_CALLS = 0


class DropoutFn(torch.autograd.Function):
    """nn.Dropout in train mode.  Counter-based RNG seeded from torch's generator seed plus a call
    counter: same distribution as torch's, not the same stream."""

    @staticmethod
    def forward(ctx, x, p: float):
        global _DROPOUT_CALLS
        require_hip(x, "dropout input")
        x = x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        _DROPOUT_CALLS += 1
        seed = (torch.initial_seed() * 0x9E3779B1 + _DROPOUT_CALLS * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF
        from .. import runtime
        _L().am_dropout_fwd(ptr(x), ptr(y), ptr(mask), x.numel(), float(p), seed, ptr(runtime.step_counter(x.device)), stream())
        ctx.p = p
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _L().am_dropout_bwd(ptr(dy), ptr(mask), ptr(dx), dy.numel(), float(ctx.p), stream())
        return dx, None


def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    return arr


class GateCombine(torch.autograd.Function):
    """(weights, combined) = gate(logits, processed...) -- gating_network.py:149-166.  `logits` are the
    (already noised, if any) gate logits; top-k masking, temperature softmax or sigmoid-normalise and the
    softmax-weighted expert combine run in one kernel."""

    @staticmethod
    def forward(ctx, logits, temperature: float, use_softmax: bool, top_k: int, *processed):
        require_hip(logits, "gate logits")
        logits = logits.contiguous()
        processed = [p.contiguous() for p in processed]
        B, E = logits.shape
        D = processed[0].shape[1]
        weights = torch.empty((B, E), dtype=torch.float32, device=logits.device)
        combined = torch.empty((B, D), dtype=torch.float32, device=logits.device)
        _L().am_gate_combine_fwd(ptr(logits), _ptr_array(processed), E, D, float(temperature), int(use_softmax), int(top_k),
                                 ptr(weights), ptr(combined), B, D, stream())
        ctx.meta = (float(temperature), int(use_softmax), int(top_k), B, E, D)
        ctx.save_for_backward(logits, *processed)
        return weights, combined

    @staticmethod
    def backward(ctx, dweights, dcombined):
        logits, *processed = ctx.saved_tensors
        temperature, use_softmax, top_k, B, E, D = ctx.meta
        dcombined = (dcombined if dcombined is not None else torch.zeros((B, D), device=logits.device)).contiguous()
        dweights = dweights.contiguous() if dweights is not None else None
        dlogits = torch.empty_like(logits)
        dproc = [torch.empty_like(p) for p in processed]
        _L().am_gate_combine_bwd(ptr(logits), _ptr_array(processed), E, D, temperature, use_softmax, top_k, ptr(dcombined),
                                 ptr(dweights), ptr(dlogits), _ptr_array(dproc), B, D, stream())
        return (dlogits, None, None, None, *dproc)


class GatingLosses(torch.autograd.Function):
    """compute_gating_losses (reference training/train_gating_network.py:21-76) and its gradient in ONE launch.
    Returns (total, parts) with parts = [ade, fde, speed, smoothness, load_balancing, entropy]; gradients flow from `total`
    only -- the parts are reporting values (non-differentiable), as in the reference's training loop."""

    @staticmethod
    def forward(ctx, wp, twp, spd, tspd, w, coef, use_lb: bool, use_ent: bool):
        require_hip(wp, "waypoints")
        wp, twp, w = wp.contiguous().float(), twp.contiguous().float(), w.contiguous().float()
        B, T = wp.shape[0], wp.shape[1]
        E = w.shape[1]
        S = 0
        if spd is not None:
            spd = spd.float()
            tspd = tspd.float()
            assert spd.stride(1) == 1 and tspd.stride(1) == 1 and spd.shape == tspd.shape
            S = spd.shape[1]
        total = torch.empty((), dtype=torch.float32, device=wp.device)
        parts = torch.empty(6, dtype=torch.float32, device=wp.device)
        g_wp = torch.empty_like(wp)
        g_w = torch.empty_like(w)
        g_spd = torch.empty((B, S), dtype=torch.float32, device=wp.device) if S else None
        c = (ctypes.c_float * 6)(*[float(v) for v in coef])
        _L().am_gating_losses(ptr(wp), ptr(twp), B, T, ptr(spd) if S else None, ptr(tspd) if S else None,
                              spd.stride(0) if S else 0, tspd.stride(0) if S else 0, S, ptr(w), E, c, int(use_lb), int(use_ent),
                              ptr(total), ptr(parts), ptr(g_wp), ptr(g_spd) if S else None, ptr(g_w), stream())
        ctx.save_for_backward(g_wp, g_spd, g_w)
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, s, _dparts):
        g_wp, g_spd, g_w = ctx.saved_tensors  # d total / d input; s = d objective / d total
        return g_wp * s, None, (g_spd * s if g_spd is not None else None), None, g_w * s, None, None, None
