"""BDDDetectionExpert -- drop-in for models/experts/bdd_detection_expert.py:4-31."""
import torch.nn as nn

from ... import runtime
from ...hip import conv as hconv
from ...hip import ops as hops
from .._nn import Conv2d, conv_bn_act
from .resnet import Trunk, load_pretrained_


class BDDDetectionExpert(nn.Module):
    def __init__(self, num_classes=10, pretrained_backbone=True):
        super().__init__()
        self.num_classes = num_classes
        self.backbone = Trunk()
        load_pretrained_(self.backbone, pretrained_backbone)
        self.head = nn.Sequential(Conv2d(512, 256, 3, padding=1), nn.ReLU(), Conv2d(256, num_classes + 4, 1))

    def features_nhwc(self, x_nhwc):
        f = self.backbone(x_nhwc)
        h = conv_bn_act(f, self.head[0], None, relu=True)
        return conv_bn_act(h, self.head[2], None, relu=False)  # [B,h,w,ld] raw head output

    def forward(self, x, nhwc_input=None):
        """x: [B,3,H,W] fp32 NCHW.  Returns NCHW fp32 channel slices, as the reference."""
        if nhwc_input is None:
            runtime.begin_step(x.device)
        xin = nhwc_input if nhwc_input is not None else hops.image_to_nhwc(x, runtime.compute_dtype())
        out = hops.NhwcToNchw.apply(self.features_nhwc(xin), self.num_classes + 4, runtime.loss_scale())
        if nhwc_input is None:
            hconv.flush_bn_counters()
        return {"class_logits": out[:, : self.num_classes, :, :], "bbox_deltas": out[:, self.num_classes:, :, :]}

    def predict(self, x):
        o = self.forward(x)
        return {"class_probs": o["class_logits"].softmax(dim=1), "bbox_deltas": o["bbox_deltas"].sigmoid()}
