"""BDDSegmentationExpert -- drop-in for models/experts/bdd_segmentation_expert.py:5-23."""
import torch.nn as nn

from ... import runtime
from ...hip import conv as hconv
from ...hip import ops as hops
from .._nn import Conv2d, conv_bn_act
from .resnet import Trunk, load_pretrained_


class _DenseExpert(nn.Module):
    """ResNet-18 trunk -> Conv3x3(512,256)+ReLU -> Conv1x1(256,C) -> bilinear upsample to the input size."""

    def __init__(self, num_classes, pretrained_backbone):
        super().__init__()
        self.num_classes = num_classes
        self.backbone = Trunk()
        load_pretrained_(self.backbone, pretrained_backbone)
        self.decoder = nn.Sequential(Conv2d(512, 256, 3, padding=1), nn.ReLU(), Conv2d(256, num_classes, 1))

    def lowres_nhwc(self, x_nhwc):
        f = self.backbone(x_nhwc)
        h = conv_bn_act(f, self.decoder[0], None, relu=True)
        return conv_bn_act(h, self.decoder[2], None, relu=False)  # [B,h,w,ld]

    def pooled_logits(self, x, nhwc_input=None):
        """AdaptiveAvgPool2d(1)(forward(x)) -> [B,C] without materialising the full-resolution logits, plus the
        low-resolution NHWC logits it was computed from (exact up to fp rounding)."""
        xin = nhwc_input if nhwc_input is not None else hops.image_to_nhwc(x, runtime.compute_dtype())
        low = self.lowres_nhwc(xin)
        return hops.UpsampleGap.apply(low, self.num_classes, x.shape[-2], x.shape[-1], runtime.loss_scale()), low

    def pixel_ce_loss(self, x, target, ignore_index=255):
        """nn.CrossEntropyLoss(ignore_index)(self(x), target) (train_bdd100k_ddp.py:89-100) without writing the full-resolution
        logits: upsample + cross entropy + their backward run on the low-resolution logits (hops.UpsampleCrossEntropy).
        Class counts without that kernel take forward() + CrossEntropy2d."""
        if not hops.UpsampleCrossEntropy.supported(self.num_classes):
            return hops.CrossEntropy2d.apply(self(x), target, ignore_index)
        runtime.begin_step(x.device)
        low = self.lowres_nhwc(hops.image_to_nhwc(x, runtime.compute_dtype()))
        hconv.flush_bn_counters()
        return hops.UpsampleCrossEntropy.apply(low, target, self.num_classes, x.shape[-2], x.shape[-1], ignore_index, runtime.loss_scale())

    def forward(self, x, nhwc_input=None):
        if nhwc_input is None:
            runtime.begin_step(x.device)
        xin = nhwc_input if nhwc_input is not None else hops.image_to_nhwc(x, runtime.compute_dtype())
        low = self.lowres_nhwc(xin)
        if nhwc_input is None:
            hconv.flush_bn_counters()
        return hops.BilinearUp.apply(low, self.num_classes, x.shape[-2], x.shape[-1], runtime.loss_scale())


class BDDSegmentationExpert(_DenseExpert):
    def __init__(self, num_classes=19, pretrained_backbone=True):
        super().__init__(num_classes, pretrained_backbone)
