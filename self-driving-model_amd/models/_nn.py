"""Parameter containers with HIP forwards.  They subclass the torch.nn layer types so constructor
signatures, default initialisation and state_dict keys are the reference's (checkpoints load
strict=True either way), but every forward runs libautomoe_hip.so kernels; none calls torch.nn.functional.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import runtime
from ..hip import conv as hconv
from ..hip import ops as hops


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters + the gather-GEMM description.  Used through conv_bn_act(), not called directly."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(cin, cout, kernel_size, stride, padding, bias=bias)
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.spec = hconv.ConvSpec(cin, cout, k, stride, padding, first=(cin == 3))
        self._packed = hconv.PackedWeights()

    def forward(self, x):  # pragma: no cover - guarded
        raise RuntimeError("Conv2d is driven through conv_bn_act() on NHWC activations")


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):  # pragma: no cover - guarded
        raise RuntimeError("BatchNorm2d is fused into conv_bn_act()")


def conv_bn_act(x: torch.Tensor, conv: Conv2d, bn: Optional[BatchNorm2d], relu: bool,
                residual: Optional[torch.Tensor] = None, give_residual_grad: bool = False, take_residual_grad: bool = False,
                pool: bool = False) -> torch.Tensor:
    """act(BN(conv(x)) + residual) on NHWC activations; BN mode follows bn.training.  give / take_residual_grad, pool (the call
    returns MaxPool2d(3,2,1) of the activation when the fused pass applies -- check the output shape): see hconv._Cfg."""
    cfg = hconv._Cfg(conv.spec, conv._packed, bn, relu, runtime.loss_scale())
    cfg.give_res_grad, cfg.take_res_grad, cfg.pool = give_residual_grad, take_residual_grad, pool
    training = bn.training if bn is not None else False
    return hconv.conv_bn_act(x, conv.weight, conv.bias, bn, relu, residual, cfg, training)


class Linear(nn.Linear):
    def forward(self, x, relu: bool = False):
        return hops.LinearAct.apply(x, self.weight, self.bias, relu)


class RowLinear(nn.Linear):
    """nn.Linear applied to MANY rows (B*Q query rows of the NuScenes decoder, models/experts/nuscenes_expert.py:139-150).
    The batch-sized GEMV kernels behind `Linear` re-read the weight per 8 rows; this one runs the rows through the
    gather-GEMM as a 1x1 convolution over an NHWC view [1, rows, 1, K] in fp32 (exact-fp32 MFMA path), with bias (+ReLU)
    in the epilogue, and the conv dgrad / wgrad kernels in backward.  Parameters keep nn.Linear's names and shapes."""

    def __init__(self, cin: int, cout: int):
        super().__init__(cin, cout)
        assert (cin * 4) % 64 == 0, "RowLinear needs K to be a multiple of 16 floats (64-byte gather runs)"
        self.spec = hconv.ConvSpec(cin, cout, 1, 1, 0)
        self._packed = hconv.PackedWeights()

    def forward(self, x, relu: bool = False):
        lead = x.shape[:-1]
        rows = x.reshape(1, -1, 1, self.in_features).float().contiguous()
        cfg = hconv._Cfg(self.spec, self._packed, None, relu, 1.0)  # fp32 rows: no loss scaling inside
        y = hconv.conv_bn_act(rows, self.weight.view(self.out_features, self.in_features, 1, 1), self.bias, None, relu, None, cfg,
                              False)
        return y[..., : self.out_features].reshape(*lead, self.out_features)


class ReLU(nn.ReLU):
    """Placeholder kept for state_dict index parity; MLPSequential fuses it into the preceding Linear."""

    def forward(self, x):  # pragma: no cover - guarded
        raise RuntimeError("ReLU must follow a Linear inside an MLPSequential (it is fused into it)")


class Dropout(nn.Dropout):
    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        return hops.DropoutFn.apply(x, self.p)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return hops.LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class GlobalAvgPoolNCHW(nn.AdaptiveAvgPool2d):
    """nn.AdaptiveAvgPool2d((1,1)) + nn.Flatten on an NCHW fp32 tensor -> [B,C]."""

    def __init__(self):
        super().__init__((1, 1))

    def forward(self, x):
        return hops.GapPlane.apply(x)


class Flatten(nn.Flatten):
    def forward(self, x):
        return x if x.dim() == 2 else x.flatten(1)


class MLPSequential(nn.Sequential):
    """nn.Sequential whose Linear -> ReLU pairs run as one fused kernel."""

    def forward(self, x, start: int = 0):
        """`start` skips leading modules (the fused upsample+pool path enters the extractor MLP after its pooling)."""
        mods = list(self)
        i = start
        while i < len(mods):
            m = mods[i]
            if isinstance(m, Linear):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = m(x, relu=fuse)
                i += 2 if fuse else 1
            else:
                x = m(x)
                i += 1
        return x


# ---- grouped launches for the parallel branches of the MoE tail (hip/ops.py GroupedLinear / GroupedLayerNorm) --------------
from ..hip.lib import AM_TAIL_MAX_GROUP as _GROUP_MAX  # noqa: E402  (include/automoe_hip.h: members of one grouped launch)


def grouped_linear(layers, inputs, relu, dropouts=None):
    """[act(layer_i(x_i))] for independent nn.Linear layers in ONE launch; `dropouts` (optional, per layer: a Dropout module or
    None) is the Dropout that follows Linear -> ReLU in the reference's Sequential, fused into the same epilogue when it is
    active (training, p > 0)."""
    layers, inputs = list(layers), list(inputs)
    dropouts = list(dropouts) if dropouts else [None] * len(layers)
    out = []
    for lo in range(0, len(layers), _GROUP_MAX):  # more members than one launch table holds (8 experts + context): several launches
        spec, args = [], []
        for layer, x, dr in zip(layers[lo:lo + _GROUP_MAX], inputs[lo:lo + _GROUP_MAX], dropouts[lo:lo + _GROUP_MAX]):
            p = float(dr.p) if (dr is not None and dr.training and dr.p > 0.0) else 0.0
            spec.append((bool(relu), p))
            args += [x, layer.weight, layer.bias]
        out += list(hops.GroupedLinear.apply(spec, *args))
    return out


def grouped_layernorm(norms, inputs):
    norms, inputs = list(norms), list(inputs)
    out = []
    for lo in range(0, len(norms), _GROUP_MAX):
        args = []
        for ln, x in zip(norms[lo:lo + _GROUP_MAX], inputs[lo:lo + _GROUP_MAX]):
            args += [x, ln.weight, ln.bias]
        out += list(hops.GroupedLayerNorm.apply([ln.eps for ln in norms[lo:lo + _GROUP_MAX]], *args))
    return out


def mlp5(seq):
    """(Linear, Dropout, Linear, LayerNorm) of a reference `Linear, ReLU, Dropout, Linear, LayerNorm` tail of a Sequential."""
    mods = list(seq)[-5:]
    assert isinstance(mods[0], Linear) and isinstance(mods[2], nn.Dropout) and isinstance(mods[3], Linear) and isinstance(mods[4], nn.LayerNorm)
    return mods[0], mods[2], mods[3], mods[4]
