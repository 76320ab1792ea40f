"""Host side of the conv gather-GEMM: geometry structs, weight packing and the autograd node
conv (+bias) (+BatchNorm) (+residual) (+ReLU) over NHWC activations.

Internal activation layout: torch tensors [B, H, W, C] (channels contiguous) of the compute dtype
(torch.float16 or torch.float32).  3-channel images are padded to 16 bytes per pixel.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from . import lib as _lib
from .lib import AM_F16, AM_F32, AM_STATS_REPLICAS, ConvGeom


def _L():
    return _lib.get()


def _runtime():
    from .. import runtime
    return runtime


def _grad_ready(p) -> bool:
    """Parameter with a preallocated, dense fp32 ``.grad`` on the same device (FusedAdamW's flat buffer views)."""
    g = getattr(p, "grad", None)
    return g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == p.device


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float16:
        return AM_F16
    if dtype == torch.float32:
        return AM_F32
    raise ValueError(f"unsupported compute dtype {dtype}")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def require_hip(t: torch.Tensor, what: str = "tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{what} is on {t.device}: the AutoMoE HIP path only runs on an MI355X (HIP) device; "
                           f"there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only)")


S2D_CH = 16  # channels of the space-to-depth(2) image handed to the 3-channel first layers (12 used)


def first_layer_taps(s: "ConvSpec"):
    """A stride-2 KxK/pad-p conv on the image == a stride-1 TAPSxTAPS conv on the space-to-depth(2) image.
    Returns (off0, taps): s2d row/col offsets of the first tap and the number of taps per axis."""
    assert s.stride == 2 and s.cin <= 4
    off0 = -((s.pad + 1) // 2)
    taps = (s.k - 1 - s.pad) // 2 - off0 + 1
    assert 1 <= taps <= 4, "first-layer kernel larger than 8x8 is not supported"
    return off0, taps


_FIRST_INDEX_CACHE = {}


def _first_index(s: "ConvSpec", device):
    """LongTensor [taps*4*16] of flat indices into w[n] (= [cin,k,k] flattened), -1 for structural zeros:
    packed position (i, j, (py*2+px)*cin + c) <- w[n, c, 2*(off0+i)+py+pad, 2*(off0+j)+px+pad]."""
    key = (s, str(device))
    if key not in _FIRST_INDEX_CACHE:
        import numpy as np
        off0, taps = first_layer_taps(s)
        idx = -np.ones((taps, 4, S2D_CH), dtype=np.int64)
        for i in range(taps):
            for j in range(taps):
                for py in range(2):
                    for px in range(2):
                        kh, kw = 2 * (off0 + i) + py + s.pad, 2 * (off0 + j) + px + s.pad
                        if 0 <= kh < s.k and 0 <= kw < s.k:
                            for c in range(s.cin):
                                idx[i, j, (py * 2 + px) * s.cin + c] = (c * s.k + kh) * s.k + kw
        flat = idx.reshape(-1)
        pos = np.nonzero(flat >= 0)[0]
        _FIRST_INDEX_CACHE[key] = (torch.from_numpy(flat).to(device), torch.from_numpy(pos).to(device),
                                   torch.from_numpy(flat[pos]).to(device))
    return _FIRST_INDEX_CACHE[key]


@dataclass(frozen=True)
class ConvSpec:
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    first: bool = False  # 3-channel input: taps are kernel rows, a run covers 8 pixels


def channel_ld(c: int, es: int) -> int:
    """Pixel stride (elements) of a C-channel NHWC activation: C itself when rows are 64-byte multiples (the gather
    run of the input-gradient GEMM), else padded to the next 64-byte multiple (heads with 14 / 19 / 3 / 10 / 7 / 4 channels)."""
    return c if (c * es) % 64 == 0 else ((c * es + 63) // 64) * 64 // es


def out_size(n: int, s: ConvSpec) -> int:
    return (n + 2 * s.pad - s.k) // s.stride + 1


def _geom(**kw) -> ConvGeom:
    g = ConvGeom()
    dy, dx = kw.pop("dy"), kw.pop("dx")
    for k, v in kw.items():
        setattr(g, k, int(v))
    for i, (a, b) in enumerate(zip(dy, dx)):
        g.dy[i], g.dx[i] = int(a), int(b)
    return g


def fwd_geom(s: ConvSpec, B: int, IH: int, IW: int, ldi: int, ldo: int, es: int, x_coff: int = 0, y_coff: int = 0,
             orig_hw=None) -> ConvGeom:
    OH, OW = out_size(IH, s), out_size(IW, s)
    if s.first:
        # x is the space-to-depth(2) image [B, IH, IW, 16]; OH/OW come from the ORIGINAL image size (orig_hw)
        off0, taps = first_layer_taps(s)
        assert ldi == S2D_CH and orig_hw is not None
        OH, OW = out_size(orig_hw[0], s), out_size(orig_hw[1], s)
        return _geom(B=B, MH=OH, MW=OW, IH=IH, IW=IW, ldi=ldi, x_coff=0, OH=OH, OW=OW, ldo=ldo, y_coff=y_coff,
                     oys=1, oy0=0, oxs=1, ox0=0, iys=1, ixs=1, ntaps=taps, krun=4 * S2D_CH, pix_shift=4, N=s.cout,
                     dy=[off0 + i for i in range(taps)], dx=[off0] * taps)
    assert (s.cin * es) % 64 == 0, f"Cin={s.cin} not a multiple of {64 // es}"
    taps = [(kh - s.pad, kw - s.pad) for kh in range(s.k) for kw in range(s.k)]
    return _geom(B=B, MH=OH, MW=OW, IH=IH, IW=IW, ldi=ldi, x_coff=x_coff, OH=OH, OW=OW, ldo=ldo, y_coff=y_coff,
                 oys=1, oy0=0, oxs=1, ox0=0, iys=s.stride, ixs=s.stride, ntaps=len(taps), krun=s.cin, pix_shift=31,
                 N=s.cout, dy=[t[0] for t in taps], dx=[t[1] for t in taps])


def pack_fwd(w: torch.Tensor, s: ConvSpec, dtype: torch.dtype) -> torch.Tensor:
    """OIHW fp32 -> [npad][taps*krun] `dtype` (k contiguous)."""
    O = w.shape[0]
    npad = _L().am_conv_npad(O)
    if s.first:
        idx = _first_index(s, w.device)[0]
        flat = torch.cat([w.detach().reshape(O, -1), w.new_zeros(O, 1)], dim=1)  # last column = structural zero
        wt = flat[:, torch.where(idx < 0, torch.full_like(idx, flat.shape[1] - 1), idx)]
    else:
        wt = w.detach().permute(0, 2, 3, 1)  # [O, kh, kw, I]
    out = torch.zeros(npad, wt[0].numel(), dtype=dtype, device=w.device)
    out[:O] = wt.reshape(O, -1).to(dtype)
    return out


def unpack_wgrad(dwp: torch.Tensor, s: ConvSpec, dtype: torch.dtype) -> torch.Tensor:
    """packed fp32 [O][taps*krun] -> OIHW fp32 gradient."""
    O = s.cout
    if s.first:
        _, pos, tgt = _first_index(s, dwp.device)  # packed positions that hold a weight, and where each one belongs
        g = torch.zeros(O, s.cin * s.k * s.k, dtype=torch.float32, device=dwp.device)
        g.index_copy_(1, tgt, dwp[:O].index_select(1, pos))  # every weight appears exactly once: no accumulation
        return g.reshape(O, s.cin, s.k, s.k)
    g = dwp[:O].reshape(O, s.k, s.k, s.cin)
    return g.permute(0, 3, 1, 2).contiguous()


def dgrad_plans(s: ConvSpec, B: int, IH: int, IW: int, ld_dx: int, ld_dy: int, es: int):
    """[(geom, tap list [(kh,kw)])] : one gather-GEMM per output-parity class of the input gradient.
    dX[n,y,x,ci] = sum_{kh,kw,co} dY[n,(y+p-kh)/s,(x+p-kw)/s,co] * W[co,ci,kh,kw] over taps whose division is exact."""
    assert not s.first
    OH, OW = out_size(IH, s), out_size(IW, s)
    assert (ld_dy * es) % 64 == 0, f"dY pixel stride {ld_dy}: dgrad run must be a multiple of 64 bytes"
    plans = []
    st = s.stride
    for py in range(st):
        for px in range(st):
            MH, MW = (IH - py + st - 1) // st, (IW - px + st - 1) // st
            if MH <= 0 or MW <= 0:
                continue
            taps = [(kh, kw) for kh in range(s.k) for kw in range(s.k)
                    if (py + s.pad - kh) % st == 0 and (px + s.pad - kw) % st == 0]
            # ascending (dy, dx): the input gradient of a 3x3 / stride-1 / pad-1 layer is then, tap for tap, the geometry of a
            # forward 3x3 convolution and takes the kernels specialised for it (layer1: the weights-in-registers kernel)
            taps.sort(key=lambda t: ((py + s.pad - t[0]) // st, (px + s.pad - t[1]) // st))
            g = _geom(B=B, MH=MH, MW=MW, IH=OH, IW=OW, ldi=ld_dy, x_coff=0, OH=IH, OW=IW, ldo=ld_dx, y_coff=0,
                      oys=st, oy0=py, oxs=st, ox0=px, iys=1, ixs=1, ntaps=len(taps), krun=ld_dy, pix_shift=31,
                      N=s.cin, dy=[(py + s.pad - kh) // st for kh, _ in taps], dx=[(px + s.pad - kw) // st for _, kw in taps])
            plans.append((g, taps))
    return plans


FOLD_EVAL_BN = os.environ.get("AM_FOLD_EVAL_BN", "1") != "0"  # eval-mode BatchNorm folded into the conv weights / bias (f16 inference)
FUSE_S2_DGRAD = os.environ.get("AM_FUSE_S2_DGRAD", "1") != "0"  # tests flip this to compare the one-launch stride-2 input gradient with the four parity-class launches

# kernel row/column index a dX pixel of parity p takes from the dY pixel at offset d (3x3, stride 2, pad 1): iy = 2*oy - 1 + kh
_S2_TAP = {(0, 0): 1, (1, 0): 2, (1, 1): 0}  # (parity, dY offset) -> kh; parity 0 never reads offset 1


def dgrad_s2_plan(s: ConvSpec, B: int, IH: int, IW: int, ld_dx: int, ld_dy: int, es: int):
    """Input gradient of a 3x3 / stride-2 / pad-1 convolution as ONE gather-GEMM (include/automoe_hip.h, `osplit`): row =
    the 2x2 block of dX pixels (2y.., 2x..), columns (py, px, ci), taps = the 2x2 dY neighbourhood (y+dy, x+dx).  None when
    the layer / tensor layout is not covered (the caller then runs the parity-class plans)."""
    if not (s.k == 3 and s.stride == 2 and s.pad == 1 and not s.first and es == 2):
        return None
    if IH % 2 or IW % 2 or ld_dx != s.cin or (2 * s.cin) % 8 or (ld_dy * es) % 64:
        return None
    OH, OW = out_size(IH, s), out_size(IW, s)
    return _geom(B=B, MH=IH // 2, MW=IW // 2, IH=OH, IW=OW, ldi=ld_dy, x_coff=0, OH=IH, OW=IW, ldo=ld_dx, y_coff=0,
                 oys=2, oy0=0, oxs=2, ox0=0, iys=1, ixs=1, ntaps=4, krun=ld_dy, pix_shift=31, N=4 * s.cin,
                 dy=[0, 0, 1, 1], dx=[0, 1, 0, 1], osplit=2 * s.cin, osplit_stride=IW * ld_dx)


def pack_dgrad_s2(w: torch.Tensor, dtype: torch.dtype, ld_dy: int) -> torch.Tensor:
    """OIHW -> [npad(4*I)][4*ld_dy]: row (py, px, ci), column (dy, dx, co); zero where the class does not use the tap."""
    O, I = w.shape[0], w.shape[1]
    npad = _L().am_conv_npad(4 * I)
    out = torch.zeros(npad, 4 * ld_dy, dtype=dtype, device=w.device)
    wt = w.detach().permute(1, 2, 3, 0)  # [I, kh, kw, O]
    for py in range(2):
        for px in range(2):
            for dy in range(2):
                for dx in range(2):
                    kh, kw = _S2_TAP.get((py, dy)), _S2_TAP.get((px, dx))
                    if kh is None or kw is None:
                        continue
                    r0, c0 = (py * 2 + px) * I, (dy * 2 + dx) * ld_dy
                    out[r0:r0 + I, c0:c0 + O] = wt[:, kh, kw, :].to(dtype)
    return out


def pack_dgrad(w: torch.Tensor, taps, dtype: torch.dtype, ld_dy: int) -> Optional[torch.Tensor]:
    """OIHW -> [npad(I)][len(taps)*ld_dy] for one parity class (O zero-padded to the dY pixel stride)."""
    if not taps:
        return None
    O, I = w.shape[0], w.shape[1]
    npad = _L().am_conv_npad(I)
    wt = w.detach().permute(1, 2, 3, 0)  # [I, kh, kw, O]
    sel = torch.stack([wt[:, kh, kw, :] for kh, kw in taps], dim=1)  # [I, T, O]
    if ld_dy != O:
        sel = torch.nn.functional.pad(sel, (0, ld_dy - O))
    out = torch.zeros(npad, sel[0].numel(), dtype=dtype, device=w.device)
    out[:I] = sel.reshape(I, -1).to(dtype)
    return out


class PackedWeights:
    """Per-conv cache of packed operands (forward layout + one dgrad layout per parity class), invalidated when the fp32
    master weight is updated.  Every layout is a fixed gather of the OIHW weight, so after an optimizer step ALL of them
    are rebuilt by one am_gather_cast launch into one flat buffer (the per-layout index maps are derived once by running
    pack_fwd / pack_dgrad on an index-valued probe weight)."""

    _ALIGN = 512  # elements: keeps every layout 1 KiB aligned inside the flat buffer

    def __init__(self):
        self.key = None
        self.layouts = {}   # name -> (offset, shape)
        self.maps = []      # int32 index maps (CPU), in offset order
        self.idx_dev = None  # concatenated maps on the device
        self.total = 0
        self.flat = None
        self.fresh = False
        self.fold = None    # (key, folded fp32 weight, folded bias, its PackedWeights): eval-mode BatchNorm folded into the conv

    def _probe_map(self, w: torch.Tensor, builder):
        probe = torch.arange(1, w.numel() + 1, dtype=torch.float32).reshape(w.shape)  # exact in fp32 below 2^24 elements
        assert w.numel() < (1 << 24)
        packed = builder(probe)
        return None if packed is None else (packed.to(torch.int64) - 1).to(torch.int32)  # structural zeros -> -1

    def _layout(self, name, w: torch.Tensor, dtype, builder):
        key = (w._version, w.data_ptr(), dtype, _runtime().weight_epoch() if w.requires_grad else -1)
        if key != self.key:
            self.key, self.fresh = key, False
        if name not in self.layouts:
            m = self._probe_map(w, builder)
            if m is None:
                self.layouts[name] = None
            else:
                n = m.numel()
                padded = -(-n // self._ALIGN) * self._ALIGN
                self.layouts[name] = (self.total, tuple(m.shape))
                self.maps.append(torch.nn.functional.pad(m.reshape(-1), (0, padded - n), value=-1))
                self.total += padded
                self.idx_dev = torch.cat(self.maps).to(w.device)
            self.fresh = False
        if self.layouts[name] is None:
            return None
        if not self.fresh:
            # a new buffer per rebuild: earlier launches (and a captured graph's earlier nodes) may still read the old one
            self.flat = torch.empty(self.total, dtype=dtype, device=w.device)
            _L().am_gather_cast(dt_code(dtype), ptr(w.detach()), ptr(self.idx_dev), ptr(self.flat), self.total, stream())
            self.fresh = True
        off, shape = self.layouts[name]
        n = 1
        for d in shape:
            n *= d
        return self.flat[off:off + n].view(shape)

    def get_fwd(self, w: torch.Tensor, s: ConvSpec, dtype: torch.dtype) -> torch.Tensor:
        return self._layout("fwd", w, dtype, lambda p: pack_fwd(p, s, torch.float32))

    def get_dgrad_s2(self, w: torch.Tensor, dtype: torch.dtype, ld_dy: int) -> torch.Tensor:
        return self._layout(("dgrad_s2", ld_dy), w, dtype, lambda p: pack_dgrad_s2(p, torch.float32, ld_dy))

    def get_dgrad(self, w: torch.Tensor, s: ConvSpec, dtype: torch.dtype, idx: int, taps, ld_dy: int) -> Optional[torch.Tensor]:
        return self._layout(("dgrad", idx, tuple(taps), ld_dy), w, dtype, lambda p: pack_dgrad(p, taps, torch.float32, ld_dy))


class PackGroup:
    """Every trainable conv layer's packed operands rebuilt by ONE am_gather_cast launch per step instead of one per layer.
    The master weights of a FusedAdamW model are views into one flat fp32 buffer, so the layers' index maps, shifted by their
    parameter offsets, concatenate into one gather from that buffer into one packed buffer (each layer's PackedWeights.flat
    becomes a slice of it).  refresh() goes first in a train step (eager, or as the first node of a captured step graph): it
    rebuilds everything from the current master weights and marks the members fresh, so the forward / backward passes launch
    no per-layer re-pack.  Layers that grow a new layout later (first backward) fall back to their own launch once and are
    folded in at the next refresh; the buffers are never reallocated while the set of layouts is unchanged."""

    def __init__(self, modules, flat_master: torch.Tensor):
        self.flat_master = flat_master
        lo, hi = flat_master.data_ptr(), flat_master.data_ptr() + flat_master.numel() * 4
        self.members = []
        for m in modules:
            pw, w = getattr(m, "_packed", None), getattr(m, "weight", None)
            if isinstance(pw, PackedWeights) and isinstance(w, torch.Tensor) and w.requires_grad and lo <= w.data_ptr() < hi:
                self.members.append((pw, w))
        self.sig = None
        self.idx = self.buf = None
        self.active = []
        # retired (idx, buf) pairs stay alive for the life of the group: a hipGraph captured before a re-layout has their
        # addresses baked into its gather and conv nodes, and a replay must never write freed memory (re-layouts happen a handful
        # of times per run -- first backward, a precision switch -- so the list stays short)
        self._retired = []

    def refresh(self, dtype: torch.dtype):
        if not self.members:
            return
        active = [(pw, w) for pw, w in self.members if pw.total > 0 and pw.idx_dev is not None]
        if not active:
            return
        sig = (dtype,) + tuple((id(pw), pw.total, w.data_ptr()) for pw, w in active)
        if sig != self.sig:
            base = self.flat_master.data_ptr()
            parts, total = [], 0
            for pw, w in active:
                off = (w.data_ptr() - base) // 4
                parts.append(torch.where(pw.idx_dev >= 0, pw.idx_dev + off, pw.idx_dev))
                total += pw.total
            assert self.flat_master.numel() < (1 << 31)
            if self.buf is not None:
                self._retired.append((self.idx, self.buf))
            self.idx = torch.cat(parts).to(torch.int32)
            self.buf = torch.empty(total, dtype=dtype, device=self.flat_master.device)
            self.sig, self.active = sig, active
        _L().am_gather_cast(dt_code(dtype), ptr(self.flat_master), ptr(self.idx), ptr(self.buf), self.buf.numel(), stream())
        epoch = _runtime().weight_epoch()
        o = 0
        for pw, w in self.active:
            pw.flat = self.buf[o:o + pw.total]
            pw.key, pw.fresh = (w._version, w.data_ptr(), dtype, epoch), True
            o += pw.total


# ---------------------------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------------------------
class KernelTimer:
    """Optional per-launch timing of the conv kernels with HIP events on the launch stream (bench.py's roofline leg).
    `flops` is the ALGORITHMIC count of the launch, 2*M*K*N with the real (unpadded) K."""

    def __init__(self):
        self.records = []  # (kind, flops, ev0, ev1, kernel, algorithmic bytes)

    def summary(self, by: str = "kind"):
        """Totals per launch kind ("conv_gemm" / "conv_dgrad" / "conv_wgrad") or, with by="kernel", per kernel the library's
        dispatcher actually launched (am_conv_last_variant)."""
        torch.cuda.synchronize()
        out = {}
        for kind, flops, e0, e1, kernel, nbytes in self.records:
            # input-gradient launches of a forward kernel are listed apart: the one-launch stride-2 form executes 16/9 of its
            # algorithmic FLOPs (structural zeros in the 2x2-block weight matrix), a forward launch exactly its own
            name = kernel + " [dgrad]" if kind == "conv_dgrad" else kernel
            d = out.setdefault(kind if by == "kind" else name, {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
        return out


TIMER: Optional[KernelTimer] = None
CONV_KERNEL_NAMES = {1: "conv_ring_k<256,256,2,4>", 2: "conv_ring_k<256,128,4,2>", 3: "conv3x3_c64n64_duo_k", 6: "conv_gemm2_k", 8: "conv_gemm_k",
                     9: "conv_s2d_k", 10: "conv_s2d_pool_k", 11: "conv_ring16_k<256,256,2,4>", 12: "conv_ring16_k<256,128,4,2>",
                     16: "conv_halo_k", 18: "conv_band16_k", 19: "conv_ring16_k<128,256,2,4>"}


def _timed(kind: str, flops: float, fn, nbytes: float = 0.0):
    """`nbytes`: ALGORITHMIC bytes of the launch (input + weights + output, each once) where the caller knows them."""
    if TIMER is None:
        fn()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    TIMER.records.append((kind, flops, e0, e1, CONV_KERNEL_NAMES.get(_L().am_conv_last_variant(), "?") if kind != "conv_wgrad" else "weight gradients (wgrad_ring_k, conv_wgrad_k, conv_s2d_wgrad_k, conv_patch_wgrad_k)", nbytes))


def conv_gemm(g: ConvGeom, x, wp, bias, relu: bool, y, stats=None, k_real: Optional[int] = None, kind: str = "conv_gemm"):
    import ctypes
    flops = 2.0 * g.B * g.MH * g.MW * (k_real if k_real is not None else g.ntaps * g.krun) * g.N
    es = y.element_size()
    nbytes = float(es) * (g.B * g.IH * g.IW * g.krun + g.N * g.ntaps * g.krun + g.B * g.MH * g.MW * g.N)
    _timed(kind, flops, lambda: _L().am_conv_gemm(ctypes.byref(g), dt_code(y.dtype), ptr(x), ptr(wp), ptr(bias), int(relu),
                                                  ptr(y), ptr(stats), stream()), nbytes)


def conv_wgrad_oihw(g: ConvGeom, x, dy, scale: float, w_param, s: "ConvSpec"):
    """Weight gradient in the nn.Conv2d layout through the workspace form (am_conv_wgrad_ws: per-chunk slabs + a summing pass,
    no atomics, no staging tensor, no re-layout).  In direct mode (runtime.direct_grads) it is ADDED straight into
    ``w_param.grad`` -- FusedAdamW's flat gradient buffer -- and None is returned (autograd has nothing to accumulate);
    otherwise the fp32 OIHW gradient is returned.  NotImplemented when the geometry has no slab form (3-channel first layers)."""
    import ctypes
    L = _L()
    code = dt_code(x.dtype)
    nbytes = ctypes.c_longlong(0)
    L.am_conv_wgrad_workspace_bytes(ctypes.byref(g), code, ctypes.byref(nbytes))
    if nbytes.value <= 0 or s.first:
        return NotImplemented
    rt = _runtime()
    direct = rt.direct_grads() and _grad_ready(w_param) and tuple(w_param.shape) == (s.cout, s.cin, s.k, s.k)
    flops = 2.0 * g.B * g.MH * g.MW * s.cin * s.k * s.k * g.N

    def launch(out):
        ws = torch.empty(nbytes.value // 4, dtype=torch.float32, device=x.device)
        _timed("conv_wgrad", flops, lambda: L.am_conv_wgrad_ws(ctypes.byref(g), code, ptr(x), ptr(dy), float(scale), ptr(ws), nbytes.value,
                                                               ptr(out), s.cin, int(direct), stream()))
        return ws

    if direct:
        launch(w_param.grad)
        rt.grad_ready(w_param)
        return None
    out = torch.empty((s.cout, s.cin, s.k, s.k), dtype=torch.float32, device=x.device)
    launch(out)
    return out


def conv_wgrad(g: ConvGeom, x, dy, scale: float, dwp, k_real: Optional[int] = None):
    import ctypes
    flops = 2.0 * g.B * g.MH * g.MW * (k_real if k_real is not None else g.ntaps * g.krun) * g.N
    _timed("conv_wgrad", flops, lambda: _L().am_conv_wgrad(ctypes.byref(g), dt_code(x.dtype), ptr(x), ptr(dy), float(scale),
                                                           ptr(dwp), stream()))


PENDING_BN_COUNTERS = []
USE_WGRAD_WORKSPACE = True  # tests flip this to compare the workspace form of the weight gradient with the atomic form
FUSE_FIRST_LAYER = True  # tests flip this to compare the fused first layer with the unfused sequence


def flush_bn_counters():
    """num_batches_tracked += 1 for every BatchNorm that ran in train mode since the last flush: one multi-tensor launch
    instead of one tiny kernel per layer."""
    if PENDING_BN_COUNTERS:
        torch._foreach_add_(list(PENDING_BN_COUNTERS), 1)
        PENDING_BN_COUNTERS.clear()
    if _RES_GRAD_STASH:
        # a block-end backward parked the identity branch's gradient for conv1's backward and nothing took it: the shortcut
        # gradient of that backward was DROPPED (an interrupted backward is the only legitimate way to get here)
        import warnings
        warnings.warn(f"{len(_RES_GRAD_STASH)} residual-gradient hand-off(s) of the previous backward were never consumed "
                      f"(keys {list(_RES_GRAD_STASH)[:3]}): that backward lost its shortcut gradients")
        _RES_GRAD_STASH.clear()


def assert_residual_handoff_consumed():
    """Called by FusedAdamW.step(): an update must never be made from a backward whose block-end gradients were parked for a
    conv1 backward that did not pick them up (keys that no longer match, a conv1 routed through another path)."""
    if _RES_GRAD_STASH:
        keys = list(_RES_GRAD_STASH)[:3]
        _RES_GRAD_STASH.clear()
        raise RuntimeError(f"residual-gradient hand-off not consumed (keys {keys}): the shortcut gradient of a BasicBlock was dropped; "
                           "set AUTOMOE_MERGE_RES_GRAD=0 to let autograd accumulate it and report this")


class _Cfg:
    """Static description of one conv(+BN)(+ReLU) layer, shared by forward and backward."""

    def __init__(self, spec: ConvSpec, cache: PackedWeights, bn=None, relu=False, loss_scale=1.0, orig_hw=None):
        self.spec, self.cache, self.bn, self.relu, self.loss_scale = spec, cache, bn, relu, loss_scale
        self.orig_hw = orig_hw  # original image size when the input is the space-to-depth image (first layers)
        self.no_grad = False    # set per call by conv_bn_act(): nothing will ask this call for a gradient
        # residual-gradient hand-off inside a BasicBlock (models/experts/resnet.py): the block end (`give`) leaves the gradient of
        # its identity branch for the block's conv1 (`take`), whose input-gradient kernel adds it in its epilogue
        self.give_res_grad = False
        self.take_res_grad = False
        # ResNet stem of a trainable trunk: conv -> BN -> ReLU -> MaxPool(3,2,1) with the normalise + ReLU + pool as ONE pass over
        # the raw conv output (am_bn_relu_maxpool3x3s2_fwd); the call returns the POOLED activation
        self.pool = False


class ConvBnAct(torch.autograd.Function):
    """y = act(BN(conv(x, w) + b) + residual), NHWC.  BN in train mode uses batch statistics and
    updates the running buffers in place (torch.nn.BatchNorm2d semantics); in eval mode the running
    statistics.  Gradients: x, w, b, gamma, beta, residual."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, residual, cfg: _Cfg, training: bool):
        require_hip(x, "conv input")
        s, L = cfg.spec, _L()
        B, IH, IW, ldi = x.shape
        es = x.element_size()
        orig_hw = (cfg.orig_hw or getattr(x, "orig_hw", None)) if s.first else None
        if s.first and orig_hw is None:
            raise RuntimeError("first-layer conv expects the space-to-depth image from hip.ops.image_to_s2d()")
        OH, OW = (out_size(orig_hw[0], s), out_size(orig_hw[1], s)) if s.first else (out_size(IH, s), out_size(IW, s))
        dtype, dev = x.dtype, x.device
        cout = s.cout
        ldo = channel_ld(cout, es)
        g = fwd_geom(s, B, IH, IW, ldi, ldo, es, orig_hw=orig_hw)
        wp = cfg.cache.get_fwd(w, s, dtype)
        bn = cfg.bn
        alloc = torch.zeros if ldo != cout else torch.empty
        raw = alloc((B, OH, OW, ldo), dtype=dtype, device=dev)
        P = B * OH * OW
        if bn is None:
            conv_gemm(g, x, wp, b, cfg.relu, raw, None, k_real=s.cin * s.k * s.k)
            y, mean, rstd = raw, None, None
        else:
            use_batch = training or bn.running_mean is None
            if not use_batch and dtype == torch.float16 and FOLD_EVAL_BN and cfg.no_grad:
                # Inference (eval-mode BatchNorm, nothing wants a gradient): y = act(conv(x, w) * scale + shift [+ residual]) with
                # constant scale / shift is the same convolution with weights w * scale[n] and bias shift[n] -- no normalise
                # pass, no raw tensor.  The folded master weight is rebuilt only when a parameter or running statistic changed.
                # FusedAdamW, am_bn_finalize and graph replays write parameters / running statistics through raw pointers (no
                # tensor version bump): the runtime's weight / statistics epochs stand in for them
                key = (w._version, w.data_ptr(), gamma._version, beta._version, bn.running_mean._version, bn.running_var._version,
                       -1 if b is None else b._version, float(bn.eps), _runtime().weight_epoch(), _runtime().stats_epoch())
                if cfg.cache.fold is None or cfg.cache.fold[0] != key:
                    sc = gamma.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                    sh = beta.detach().float() - bn.running_mean.float() * sc
                    if b is not None:
                        sh = sh + b.detach().float() * sc
                    cfg.cache.fold = (key, (w.detach().float() * sc.view(-1, 1, 1, 1)).contiguous(), sh.contiguous(), PackedWeights())
                _, wf, bf, cache_f = cfg.cache.fold
                if residual is None:
                    conv_gemm(g, x, cache_f.get_fwd(wf, s, dtype), bf, cfg.relu, raw, None, k_real=s.cin * s.k * s.k)
                    return raw
                # block end: act(conv + bias + identity) in the conv epilogue (am_conv_gemm_res) -- no normalise + add pass.
                # Shapes without that epilogue fall through to the unfolded sequence below
                if tuple(residual.shape) == tuple(raw.shape) and residual.is_contiguous() and residual.dtype == dtype:
                    import ctypes
                    flops = 2.0 * g.B * g.MH * g.MW * s.cin * s.k * s.k * g.N
                    try:
                        _timed("conv_gemm", flops, lambda: L.am_conv_gemm_res(ctypes.byref(g), dt_code(dtype), ptr(x), ptr(cache_f.get_fwd(wf, s, dtype)),
                                                                              ptr(bf), ptr(residual), int(cfg.relu), ptr(raw), stream()))
                        return raw
                    except RuntimeError as e:
                        if "UNSUPPORTED" not in str(e):
                            raise
            stats = _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * cout, dev) if use_batch else None
            # Frozen first layer in train-mode BN (AutoMoE's gating stage): two light passes over the image instead of
            # conv -> raw output -> normalise pass.  Only when nothing here needs a gradient (raw output is not kept).
            fused_first = (s.first and use_batch and b is None and residual is None and cfg.relu and dtype == torch.float16
                           and not (w.requires_grad or gamma.requires_grad or beta.requires_grad) and FUSE_FIRST_LAYER)
            if fused_first:
                import ctypes
                try:
                    # statistics pass: a recompute, so it adds time but no algorithmic FLOPs to the roofline accounting
                    _timed("conv_gemm", 0.0,
                           lambda: L.am_conv_first_fused(ctypes.byref(g), AM_F16, 1, ptr(x), ptr(wp), None, None, None, ptr(stats), stream()))
                except RuntimeError as e:
                    if "UNSUPPORTED" not in str(e):
                        raise
                    fused_first = False
            if not fused_first:
                conv_gemm(g, x, wp, b, False, raw, stats, k_real=s.cin * s.k * s.k)
            mean = torch.empty(cout, dtype=torch.float32, device=dev)
            rstd = torch.empty_like(mean)
            momentum = bn.momentum if bn.momentum is not None else 0.1
            upd = use_batch and bn.track_running_stats and bn.running_mean is not None
            rmean = ptr(bn.running_mean) if (upd or not use_batch) else None
            rvar = ptr(bn.running_var) if (upd or not use_batch) else None
            if upd:
                _runtime().bump_stats_epoch()
            if upd and bn.num_batches_tracked is not None:
                PENDING_BN_COUNTERS.append(bn.num_batches_tracked)  # bumped together by flush_bn_counters()
            if fused_first:
                import ctypes
                scale = torch.empty_like(mean)
                shift = torch.empty_like(mean)
                L.am_bn_finalize(ptr(stats), AM_STATS_REPLICAS, float(P), ptr(b) if use_batch else None, ptr(gamma), ptr(beta),
                                 rmean, rvar, float(momentum), float(bn.eps), int(use_batch), ptr(scale), ptr(shift), ptr(mean),
                                 ptr(rstd), cout, stream())
                _timed("conv_gemm", 2.0 * P * s.cin * s.k * s.k * cout,
                       lambda: L.am_conv_first_fused(ctypes.byref(g), AM_F16, 2, ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(raw), None, stream()))
                return raw  # no graph: nothing requires grad
            # (a single finalize+apply launch was measured slower: every workgroup repeats the fp64 prologue -- 1738 vs 1795 img/s)
            scale = torch.empty_like(mean)
            shift = torch.empty_like(mean)
            L.am_bn_finalize(ptr(stats), AM_STATS_REPLICAS, float(P), ptr(b) if use_batch else None, ptr(gamma), ptr(beta),
                             rmean, rvar, float(momentum), float(bn.eps), int(use_batch), ptr(scale), ptr(shift), ptr(mean),
                             ptr(rstd), cout, stream())
            pooled = cfg.pool and cfg.relu and residual is None and SIGN_RELU_MASK and ldo == cout
            ctx.pool_arg = None
            if pooled:
                # normalise + ReLU + MaxPool(3,2,1) in one pass: the full-resolution activation is neither written nor re-read
                # (backward needs the arg-max and, for the ReLU mask, only the sign of raw * scale + shift)
                POH, POW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
                y = torch.empty((B, POH, POW, cout), dtype=dtype, device=dev)
                ctx.pool_arg = torch.empty((B, POH, POW, cout), dtype=torch.uint8, device=dev)
                L.am_bn_relu_maxpool3x3s2_fwd(dt_code(dtype), ptr(raw), ptr(scale), ptr(shift), ptr(y), ptr(ctx.pool_arg), B, OH, OW, cout, stream())
            else:
                y = torch.empty_like(raw)
                L.am_bn_apply(dt_code(dtype), ptr(raw), ldo, ptr(scale), ptr(shift), ptr(residual),
                              residual.shape[-1] if residual is not None else 0, int(cfg.relu), ptr(y), ldo, P, cout, stream())
            ctx.use_batch = use_batch
            # BatchNorm + ReLU without a residual: backward takes the ReLU mask from the sign of raw * scale + shift (the *_sign
            # entries) instead of reading y -- one tensor read less in each of its passes
            ctx.sign_ss = (scale, shift) if (SIGN_RELU_MASK and cfg.relu and residual is None) else None
        ctx.cfg, ctx.geom = cfg, g
        ctx.has_res = residual is not None
        ctx.give_key = (residual.data_ptr(), tuple(residual.shape)) if (cfg.give_res_grad and residual is not None) else None
        ctx.take_key = (x.data_ptr(), tuple(x.shape)) if cfg.take_res_grad else None
        ctx.bn_params = (gamma, beta)
        ctx.w_param = w  # the Parameter object itself (direct-mode weight gradients go into its .grad)
        keep_y = (cfg.relu or bn is None) and not (bn is not None and getattr(ctx, "pool_arg", None) is not None)
        ctx.save_for_backward(x, w, b, gamma, raw if bn is not None else None, y if keep_y else None, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, gamma, raw, y, mean, rstd = ctx.saved_tensors
        cfg, g, L = ctx.cfg, ctx.geom, _L()
        s = cfg.spec
        dtype, dev = x.dtype, x.device
        es = x.element_size()
        pool_sums = None
        if cfg.bn is not None and getattr(ctx, "pool_arg", None) is not None:
            # the output was the pooled activation: its gradient goes back to the conv-output resolution first -- and the same pass
            # accumulates the BatchNorm-backward sums (am_maxpool3x3s2_bwd_bn), so the reduce pass below is skipped
            dy = dy.contiguous()
            full = torch.empty_like(raw)
            pool_sums = _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * s.cout, dev)
            try:
                L.am_maxpool3x3s2_bwd_bn(dt_code(dtype), ptr(dy), ptr(ctx.pool_arg), ptr(full), raw.shape[0], raw.shape[1], raw.shape[2], raw.shape[3],
                                         ptr(raw), ptr(mean), ptr(rstd), ptr(ctx.sign_ss[0]), ptr(ctx.sign_ss[1]), ptr(pool_sums), stream())
            except RuntimeError as e:
                if "UNSUPPORTED" not in str(e):
                    raise
                pool_sums = None
                L.am_maxpool3x3s2_bwd(dt_code(dtype), ptr(dy), ptr(ctx.pool_arg), ptr(full), raw.shape[0], raw.shape[1], raw.shape[2], raw.shape[3], stream())
            dy = full
        B, OH, OW, ldo = dy.shape
        P = B * OH * OW
        cout = s.cout
        inv = 1.0 / cfg.loss_scale
        dy = dy.contiguous()
        code = dt_code(dtype)
        db = dgamma = dbeta = dres = None
        if cfg.bn is not None:
            sums = pool_sums if pool_sums is not None else _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * cout, dev)
            sign_ss = ctx.sign_ss
            if pool_sums is not None:
                pass  # filled by am_maxpool3x3s2_bwd_bn above
            elif sign_ss is not None:
                L.am_bn_bwd_reduce_sign(code, ptr(dy), ldo, ptr(raw), ldo, ptr(mean), ptr(rstd), ptr(sign_ss[0]), ptr(sign_ss[1]), ptr(sums), P,
                                        cout, stream())
            else:
                L.am_bn_bwd_reduce(code, ptr(dy), ldo, ptr(y), ldo, ptr(raw), ldo, ptr(mean), ptr(rstd), int(cfg.relu), ptr(sums), P,
                                   cout, stream())
            coef = torch.empty(3 * cout, dtype=torch.float32, device=dev)
            need_p = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
            gp, bp = ctx.bn_params
            direct = (need_p and _runtime().direct_grads() and ctx.needs_input_grad[3] and ctx.needs_input_grad[4]
                      and _grad_ready(gp) and _grad_ready(bp))
            if need_p and not direct:
                dgamma = torch.zeros(cout, dtype=torch.float32, device=dev)
                dbeta = torch.zeros(cout, dtype=torch.float32, device=dev)
            # the kernel accumulates (+=): in direct mode straight into the optimizer's gradient buffer
            L.am_bn_bwd_finalize(ptr(sums), AM_STATS_REPLICAS, float(P), ptr(gamma), ptr(rstd), inv,
                                 ptr(gp.grad) if direct else ptr(dgamma), ptr(bp.grad) if direct else ptr(dbeta), ptr(coef), cout, stream())
            if direct:
                _runtime().grad_ready(gp, bp)
            if not ctx.use_batch:
                coef[cout:].zero_()  # eval-mode BN: statistics are constants
            fused_wgrad = False
            if (s.first and not ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not ctx.has_res and dtype == torch.float16
                    and not (b is not None and ctx.needs_input_grad[2] and not ctx.use_batch)):
                # first layer: nothing consumes the gradient w.r.t. the conv output except the weight gradient -- form it inside
                # the wgrad kernel's tile load instead of writing and re-reading it (am_conv_wgrad_bn)
                import ctypes
                ktot = g.ntaps * g.krun
                dwp = torch.zeros(cout, ktot, dtype=torch.float32, device=dev)
                try:
                    if sign_ss is not None:
                        _timed("conv_wgrad", 2.0 * P * s.cin * s.k * s.k * cout,
                               lambda: L.am_conv_wgrad_bn_sign(ctypes.byref(g), code, ptr(x), ptr(dy), ptr(raw), ptr(mean), ptr(rstd), ptr(coef),
                                                               ptr(sign_ss[0]), ptr(sign_ss[1]), inv, ptr(dwp), stream()))
                    else:
                        _timed("conv_wgrad", 2.0 * P * s.cin * s.k * s.k * cout,
                               lambda: L.am_conv_wgrad_bn(ctypes.byref(g), code, ptr(x), ptr(dy), ptr(y) if cfg.relu else None, ptr(raw), ptr(mean),
                                                          ptr(rstd), ptr(coef), int(cfg.relu), inv, ptr(dwp), stream()))
                    fused_wgrad = True
                except RuntimeError as e:
                    if "UNSUPPORTED" not in str(e):
                        raise
            if fused_wgrad:
                dw = unpack_wgrad(dwp, s, dtype)
                if b is not None and ctx.needs_input_grad[2]:
                    db = torch.zeros_like(b)  # conv bias feeding a train-mode BN has an exactly zero gradient
                return None, dw, db, dgamma, dbeta, None, None, None
            dz = torch.empty_like(dy)
            dres_t = torch.empty_like(dy) if (ctx.has_res and ctx.needs_input_grad[5]) else None
            if sign_ss is not None:
                L.am_bn_bwd_apply_sign(code, ptr(dy), ldo, ptr(raw), ldo, ptr(mean), ptr(rstd), ptr(coef), ptr(sign_ss[0]), ptr(sign_ss[1]),
                                       ptr(dz), ldo, P, cout, stream())
            else:
                L.am_bn_bwd_apply(code, ptr(dy), ldo, ptr(y), ldo, ptr(raw), ldo, ptr(mean), ptr(rstd), ptr(coef), int(cfg.relu),
                                  ptr(dz), ldo, ptr(dres_t), ldo, P, cout, stream())
            dres = dres_t
            if dres is not None and ctx.give_key is not None and MERGE_RESIDUAL_GRAD:
                _RES_GRAD_STASH[ctx.give_key] = dres  # conv1 of this block adds it to its input gradient (same tensor: the block input)
                dres = None
            if b is not None and ctx.needs_input_grad[2]:
                # conv bias feeding a train-mode BN has an exactly zero gradient; in eval mode it is colsum(dz)
                db = torch.zeros_like(b)
                if not ctx.use_batch:
                    L.am_bias_relu_bwd(code, ptr(dz), ldo, None, 0, 0, None, 0, ptr(db), inv, P, ldo, cout, stream())
        else:
            need_b = b is not None and ctx.needs_input_grad[2]
            if cfg.relu or need_b:
                db = torch.zeros_like(b) if need_b else None
                dz = torch.empty_like(dy) if cfg.relu else dy
                L.am_bias_relu_bwd(code, ptr(dy), ldo, ptr(y), ldo, int(cfg.relu), ptr(dz) if cfg.relu else None, ldo, ptr(db),
                                   inv, P, ldo, cout, stream())
            else:
                dz = dy
        dx = dw = None
        if ctx.needs_input_grad[0]:
            _, IH, IW, ldi = x.shape
            dx = torch.empty_like(x)
            g2 = dgrad_s2_plan(s, B, IH, IW, ldi, ldo, es) if FUSE_S2_DGRAD else None
            if g2 is not None:
                try:
                    conv_gemm(g2, dz, cfg.cache.get_dgrad_s2(w, dtype, ldo), None, False, dx, None, k_real=2.25 * s.cout, kind="conv_dgrad")
                except RuntimeError as e:
                    if "UNSUPPORTED" not in str(e):
                        raise
                    g2 = None
            if g2 is None:
                plans = dgrad_plans(s, B, IH, IW, ldi, ldo, es)
                pending = _RES_GRAD_STASH.pop(ctx.take_key, None) if ctx.take_key is not None else None
                for idx, (gd, taps) in enumerate(plans):
                    wd = cfg.cache.get_dgrad(w, s, dtype, idx, taps, ldo)
                    if pending is not None and len(plans) == 1 and pending.shape == dx.shape and pending.dtype == dx.dtype:
                        # dX = dgrad(dz) + (gradient of the block's identity branch) in the kernel's epilogue: no accumulate pass
                        import ctypes
                        try:
                            _timed("conv_dgrad", 2.0 * gd.B * gd.MH * gd.MW * len(taps) * s.cout * gd.N,
                                   lambda: L.am_conv_gemm_res(ctypes.byref(gd), code, ptr(dz), ptr(wd), None, ptr(pending), 0, ptr(dx), stream()))
                            pending = None
                            RES_GRAD_COUNTS["fused"] += 1
                            continue
                        except RuntimeError as e:
                            if "UNSUPPORTED" not in str(e):
                                raise
                    conv_gemm(gd, dz, wd, None, False, dx, None, k_real=len(taps) * s.cout, kind="conv_dgrad")
                if pending is not None:
                    dx.add_(pending)
                    RES_GRAD_COUNTS["added"] += 1
            elif ctx.take_key is not None and ctx.take_key in _RES_GRAD_STASH:
                dx.add_(_RES_GRAD_STASH.pop(ctx.take_key))
                RES_GRAD_COUNTS["added"] += 1
        elif ctx.take_key is not None:
            _RES_GRAD_STASH.pop(ctx.take_key, None)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad_oihw(g, x, dz, inv, ctx.w_param, s) if USE_WGRAD_WORKSPACE else NotImplemented
            if dw is NotImplemented:  # no slab form for this geometry: atomics into a packed staging tensor, then the re-layout
                ktot = g.ntaps * g.krun
                dwp = torch.zeros(cout, ktot, dtype=torch.float32, device=dev)
                conv_wgrad(g, x, dz, inv, dwp, k_real=s.cin * s.k * s.k)
                dw = unpack_wgrad(dwp, s, dtype)
        return dx, dw, db, dgamma, dbeta, dres, None, None


SIGN_RELU_MASK = os.environ.get("AUTOMOE_SIGN_RELU_MASK", "1") != "0"  # tests flip this to compare with the mask read from the activation
MERGE_RESIDUAL_GRAD = os.environ.get("AUTOMOE_MERGE_RES_GRAD", "1") != "0"  # tests flip this to compare with autograd's own accumulation of the two gradients of a block input
_RES_GRAD_STASH = {}        # (data_ptr, shape) of a block input -> gradient of the block's identity branch, until conv1's backward
RES_GRAD_COUNTS = {"fused": 0, "added": 0}  # hand-offs taken by a conv epilogue / by an in-place add (tests)


def conv_bn_act(x, w, b, bn, relu: bool, residual, cfg: _Cfg, training: bool):
    if cfg.spec.first and cfg.orig_hw is None:
        cfg.orig_hw = getattr(x, "orig_hw", None)
    gamma = bn.weight if bn is not None else None
    beta = bn.bias if bn is not None else None
    # (inside Function.forward grad mode is always off and needs_input_grad still mirrors requires_grad: decide here)
    cfg.no_grad = not torch.is_grad_enabled() or not any(t is not None and t.requires_grad for t in (x, w, b, gamma, beta, residual))
    return ConvBnAct.apply(x, w, b, gamma, beta, residual, cfg, training)


STEM_ONE_PASS = os.environ.get("AUTOMOE_STEM_ONE_PASS", "1") != "0"  # tests flip this to compare with the two-pass form


class PendingAffine:
    """A pooled stem map whose BatchNorm + ReLU is still to be applied: consumers read relu(raw * scale[c] + shift[c]) (scale >= 0;
    am_conv_first_fused mode 4 / am_bn_finalize_signed).  `materialize()` writes that tensor for consumers without a fused form."""
    __slots__ = ("raw", "scale", "shift")

    def __init__(self, raw, scale, shift):
        self.raw, self.scale, self.shift = raw, scale, shift

    def materialize(self):
        B, H, W, C = self.raw.shape
        y = torch.empty_like(self.raw)
        _L().am_bn_apply(dt_code(self.raw.dtype), ptr(self.raw), C, ptr(self.scale), ptr(self.shift), None, 0, 1, ptr(y), C, B * H * W, C, stream())
        return y


@torch.no_grad()
def fused_stem_pool(x, conv_w, bn, cfg: _Cfg, allow_pending: bool = False):
    """ResNet stem for a FROZEN trunk in train-mode BatchNorm: conv7x7/s2 -> BN(batch statistics, running stats
    updated) -> ReLU -> MaxPool(3,2,1) without the raw conv output or the normalised map ever reaching HBM.
    Round 3 (allow_pending, STEM_ONE_PASS): ONE pass over the space-to-depth image -- the conv output is pooled raw (on
    sign(gamma) * conv, so that a negative gamma pools the minimum) next to its BatchNorm sums, and the normalisation + ReLU,
    which commute with the max-pool, are left to the consumers of the pooled map: the call returns a PendingAffine.
    Otherwise two passes: a statistics-only pass and a pass whose epilogue normalises, rectifies and pools.
    Eval-mode BatchNorm (inference): the second pass alone, with scale / shift from the running statistics.
    Returns the pooled NHWC activation (or PendingAffine), or None when the case is not covered (caller runs the unfused sequence)."""
    import ctypes
    s = cfg.spec
    if not (FUSE_FIRST_LAYER and s.first and x.dtype == torch.float16 and bn is not None and s.cout == 64):
        return None
    # train-mode BatchNorm: only a frozen stem (no gradient through the fused passes); eval mode (inference: running statistics
    # are constants, no statistics pass): whenever nothing asks for a gradient
    if (conv_w.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad) and (bn.training or torch.is_grad_enabled()):
        return None
    if not bn.training and bn.running_mean is None:
        return None
    orig_hw = cfg.orig_hw or getattr(x, "orig_hw", None)
    if orig_hw is None:
        return None
    L = _L()
    B, IH, IW, ldi = x.shape
    g = fwd_geom(s, B, IH, IW, ldi, 64, 2, orig_hw=orig_hw)
    if g.ntaps != 4:
        return None
    OH, OW = g.OH, g.OW
    P = B * OH * OW
    dev = x.device
    wp = cfg.cache.get_fwd(conv_w, s, x.dtype)
    scale = torch.empty(64, dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    POH, POW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    if bn.training and STEM_ONE_PASS and allow_pending:
        # ONE pass over the image (am_conv_first_fused mode 4): sign(gamma) * conv pooled raw + its BatchNorm sums; the
        # normalisation + ReLU commute with the max-pool and are applied by the consumers of the pooled map (PendingAffine)
        stats = _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * 64, dev)
        y = torch.empty((B, POH, POW, 64), dtype=x.dtype, device=dev)
        done = True
        try:
            _timed("conv_gemm", 2.0 * P * s.cin * s.k * s.k * 64,
                   lambda: L.am_conv_first_fused(ctypes.byref(g), AM_F16, 4, ptr(x), ptr(wp), ptr(bn.weight), None, ptr(y), ptr(stats), stream()))
        except RuntimeError as e:
            if "UNSUPPORTED" not in str(e):
                raise
            done = False
        if done:
            momentum = bn.momentum if bn.momentum is not None else 0.1
            upd = bn.track_running_stats and bn.running_mean is not None
            L.am_bn_finalize_signed(ptr(stats), AM_STATS_REPLICAS, float(P), ptr(bn.weight), ptr(bn.bias),
                                    ptr(bn.running_mean) if upd else None, ptr(bn.running_var) if upd else None, float(momentum),
                                    float(bn.eps), ptr(scale), ptr(shift), 64, stream())
            if upd:
                _runtime().bump_stats_epoch()
            if upd and bn.num_batches_tracked is not None:
                PENDING_BN_COUNTERS.append(bn.num_batches_tracked)
            return PendingAffine(y, scale, shift)
    if bn.training:
        stats = _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * 64, dev)
        try:
            _timed("conv_gemm", 0.0, lambda: L.am_conv_first_fused(ctypes.byref(g), AM_F16, 1, ptr(x), ptr(wp), None, None, None,
                                                                   ptr(stats), stream()))
        except RuntimeError as e:
            if "UNSUPPORTED" not in str(e):
                raise
            return None
        momentum = bn.momentum if bn.momentum is not None else 0.1
        upd = bn.track_running_stats and bn.running_mean is not None
        L.am_bn_finalize(ptr(stats), AM_STATS_REPLICAS, float(P), None, ptr(bn.weight), ptr(bn.bias),
                         ptr(bn.running_mean) if upd else None, ptr(bn.running_var) if upd else None, float(momentum),
                         float(bn.eps), 1, ptr(scale), ptr(shift), None, None, 64, stream())
        if upd:
            _runtime().bump_stats_epoch()
        if upd and bn.num_batches_tracked is not None:
            PENDING_BN_COUNTERS.append(bn.num_batches_tracked)
    else:
        # eval mode: scale / shift from the running statistics (am_bn_finalize's eval branch), ONE pass over the image
        L.am_bn_finalize(None, AM_STATS_REPLICAS, float(P), None, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                         0.0, float(bn.eps), 0, ptr(scale), ptr(shift), None, None, 64, stream())
    y = torch.empty((B, POH, POW, 64), dtype=x.dtype, device=dev)
    try:
        _timed("conv_gemm", 2.0 * P * s.cin * s.k * s.k * 64,
               lambda: L.am_conv_first_fused(ctypes.byref(g), AM_F16, 3, ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(y), None, stream()))
    except RuntimeError as e:
        if "UNSUPPORTED" not in str(e) or bn.training:
            raise
        return None
    return y


FUSE_BLOCK_BN = True  # tests flip this to compare with the unfused sequence


@torch.no_grad()
def fused_basic_block_identity(x, conv1_w, bn1, cache1: PackedWeights, conv2_w, bn2, cache2: PackedWeights, pre=None):
    """ResNet BasicBlock C -> C (stride 1, identity shortcut; C = 64: layer1, C = 128: layer2) of a FROZEN trunk in train-mode
    BatchNorm, f16:
        y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)
    with bn1 + ReLU applied inside conv2's input staging (am_conv_gemm_prebn: the weights-in-registers kernel for C = 64, the
    halo-staged kernel for C = 128): relu(bn1(.)) is never written or re-read -- one 4-bytes-per-element HBM pass less per block.
    Both BatchNorms use batch statistics and update their running statistics exactly as the unfused sequence.  Returns None
    when the case is not covered (caller runs the unfused blocks).
    `pre` = (scale, shift): the block input is relu(x * scale + shift) with x the raw pooled stem (PendingAffine): conv1 forms it in
    its input staging as conv2 does for bn1, the block end forms it for the residual (am_bn_apply2, relu bits 0 and 1)."""
    import ctypes
    if not (FUSE_BLOCK_BN and x.dtype == torch.float16 and bn1.training and bn2.training):
        return None
    if any(t.requires_grad for t in (conv1_w, conv2_w, bn1.weight, bn1.bias, bn2.weight, bn2.bias)):
        return None
    B, H, W, C = x.shape
    if C not in (64, 128) or tuple(conv1_w.shape) != (C, C, 3, 3) or tuple(conv2_w.shape) != (C, C, 3, 3):
        return None
    L, dev = _L(), x.device
    s = ConvSpec(C, C, 3, 1, 1)
    g = fwd_geom(s, B, H, W, C, C, 2)
    P = B * H * W
    flops = 2.0 * P * C * 9 * C
    raw1 = torch.empty_like(x)
    stats1 = _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * C, dev)
    w1 = cache1.get_fwd(conv1_w, s, x.dtype)
    nbytes = 2.0 * (2 * P * C + 9 * C * C)  # algorithmic: input + output once each, weights
    if pre is not None:
        try:
            _timed("conv_gemm", flops, lambda: L.am_conv_gemm_prebn(ctypes.byref(g), AM_F16, ptr(x), ptr(pre[0]), ptr(pre[1]), ptr(w1), ptr(raw1),
                                                                  ptr(stats1), stream()), nbytes)
        except RuntimeError as e:
            if "UNSUPPORTED" not in str(e):
                raise
            return None  # (nothing has touched the BatchNorm buffers yet: the caller materialises the input and runs the plain form)
    else:
        conv_gemm(g, x, w1, None, False, raw1, stats1, k_real=9 * C)
    sc1, sh1 = _bn_finalize_nograd(bn1, stats1, P, C)
    raw2 = torch.empty_like(x)
    stats2 = _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * C, dev)
    w2 = cache2.get_fwd(conv2_w, s, x.dtype)
    try:
        _timed("conv_gemm", flops, lambda: L.am_conv_gemm_prebn(ctypes.byref(g), AM_F16, ptr(raw1), ptr(sc1), ptr(sh1), ptr(w2), ptr(raw2),
                                                              ptr(stats2), stream()), nbytes)
    except RuntimeError as e:
        if "UNSUPPORTED" not in str(e):
            raise
        y1 = torch.empty_like(x)  # small problem: apply bn1 the ordinary way
        L.am_bn_apply(AM_F16, ptr(raw1), C, ptr(sc1), ptr(sh1), None, 0, 1, ptr(y1), C, P, C, stream())
        conv_gemm(g, y1, w2, None, False, raw2, stats2, k_real=9 * C)
    sc2, sh2 = _bn_finalize_nograd(bn2, stats2, P, C)
    y = torch.empty_like(x)
    if pre is not None:
        L.am_bn_apply2(AM_F16, ptr(raw2), C, ptr(sc2), ptr(sh2), ptr(x), C, ptr(pre[0]), ptr(pre[1]), 3, ptr(y), C, P, C, stream())
    else:
        L.am_bn_apply(AM_F16, ptr(raw2), C, ptr(sc2), ptr(sh2), ptr(x), C, 1, ptr(y), C, P, C, stream())
    return y


def _conv_stats_nograd(x, w, s: ConvSpec, cache: PackedWeights):
    """Raw convolution output + BatchNorm batch statistics of a frozen layer (no autograd bookkeeping)."""
    B, IH, IW, ldi = x.shape
    ldo = channel_ld(s.cout, x.element_size())
    g = fwd_geom(s, B, IH, IW, ldi, ldo, x.element_size())
    raw = (torch.zeros if ldo != s.cout else torch.empty)((B, g.OH, g.OW, ldo), dtype=x.dtype, device=x.device)
    stats = _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * s.cout, x.device)
    conv_gemm(g, x, cache.get_fwd(w, s, x.dtype), None, False, raw, stats, k_real=s.cin * s.k * s.k)
    return raw, stats, B * g.OH * g.OW


def _bn_finalize_nograd(bn, stats, P: int, C: int):
    scale = torch.empty(C, dtype=torch.float32, device=stats.device)
    shift = torch.empty_like(scale)
    momentum = bn.momentum if bn.momentum is not None else 0.1
    upd = bn.track_running_stats and bn.running_mean is not None
    _L().am_bn_finalize(ptr(stats), AM_STATS_REPLICAS, float(P), None, ptr(bn.weight), ptr(bn.bias),
                        ptr(bn.running_mean) if upd else None, ptr(bn.running_var) if upd else None, float(momentum), float(bn.eps), 1,
                        ptr(scale), ptr(shift), None, None, C, stream())
    if upd:
        _runtime().bump_stats_epoch()
    if upd and bn.num_batches_tracked is not None:
        PENDING_BN_COUNTERS.append(bn.num_batches_tracked)
    return scale, shift


@torch.no_grad()
def fused_basic_block_down(x, blk):
    """Strided ResNet BasicBlock (3x3/s2 conv1, 1x1/s2 downsample shortcut) of a FROZEN trunk in train-mode BatchNorm:
        y = relu(bn2(conv2(relu(bn1(conv1(x))))) + bn_d(conv_d(x)))
    with the shortcut's BatchNorm applied inside the final normalise + add + ReLU pass (am_bn_apply2) instead of a pass of
    its own.  `blk` is models.experts.resnet.BasicBlock.  Returns None when the case is not covered."""
    mods = (blk.conv1, blk.bn1, blk.conv2, blk.bn2, blk.downsample[0], blk.downsample[1])
    if not (FUSE_BLOCK_BN and os.environ.get("AM_FUSE_DOWN", "1") != "0" and blk.bn1.training and blk.bn2.training and blk.downsample[1].training):
        return None
    if any(p.requires_grad for m in mods for p in m.parameters()) or any(m.bias is not None for m in (blk.conv1, blk.conv2, blk.downsample[0])):
        return None
    L = _L()
    code = dt_code(x.dtype)
    C = blk.conv1.spec.cout
    raw1, st1, P = _conv_stats_nograd(x, blk.conv1.weight, blk.conv1.spec, blk.conv1._packed)
    raw_d, st_d, _ = _conv_stats_nograd(x, blk.downsample[0].weight, blk.downsample[0].spec, blk.downsample[0]._packed)
    ld = raw1.shape[-1]
    if ld != C:
        return None
    sc1, sh1 = _bn_finalize_nograd(blk.bn1, st1, P, C)
    raw2 = st2 = None
    s2 = blk.conv2.spec
    if x.dtype == torch.float16 and s2.k == 3 and s2.stride == 1 and s2.pad == 1:
        # conv2 on relu(bn1(raw1)) formed in its input staging (am_conv_gemm_prebn): the normalised map is never written
        import ctypes
        Bn, OH, OW, _ = raw1.shape
        g2 = fwd_geom(s2, Bn, OH, OW, ld, ld, 2)
        raw2 = torch.empty_like(raw1)
        st2 = _runtime().arena_zeros(AM_STATS_REPLICAS * 2 * C, x.device)
        try:
            _timed("conv_gemm", 2.0 * P * s2.cin * 9 * C, lambda: L.am_conv_gemm_prebn(
                ctypes.byref(g2), AM_F16, ptr(raw1), ptr(sc1), ptr(sh1), ptr(blk.conv2._packed.get_fwd(blk.conv2.weight, s2, x.dtype)),
                ptr(raw2), ptr(st2), stream()), 2.0 * (2 * P * C + 9 * C * C))
        except RuntimeError as e:
            if "UNSUPPORTED" not in str(e):
                raise
            raw2 = None
    if raw2 is None:
        y1 = torch.empty_like(raw1)
        L.am_bn_apply(code, ptr(raw1), ld, ptr(sc1), ptr(sh1), None, 0, 1, ptr(y1), ld, P, C, stream())
        raw2, st2, _ = _conv_stats_nograd(y1, blk.conv2.weight, blk.conv2.spec, blk.conv2._packed)
    sc2, sh2 = _bn_finalize_nograd(blk.bn2, st2, P, C)
    sc_d, sh_d = _bn_finalize_nograd(blk.downsample[1], st_d, P, C)
    y = torch.empty_like(raw2)
    L.am_bn_apply2(code, ptr(raw2), ld, ptr(sc2), ptr(sh2), ptr(raw_d), ld, ptr(sc_d), ptr(sh_d), 1, ptr(y), ld, P, C, stream())
    return y
