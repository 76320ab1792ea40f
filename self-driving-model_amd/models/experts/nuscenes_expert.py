"""NuScenesExpert -- drop-in for models/experts/nuscenes_expert.py:96-190 (image branch).

ResNet-18 trunk + global average pool -> Linear(512, 256) -> one scene feature per image, broadcast over `num_queries`
learned query embeddings -> decoder MLP per (image, query) row -> class / box heads.  The trunk runs the same NHWC
gather-GEMM kernels as the BDD experts (so it can share the space-to-depth image inside AutoMoE); the decoder runs its
B*Q rows through the fp32 gather-GEMM (`RowLinear`).

Not built: the LiDAR branch (`use_lidar=True`: PointNet / T-Net, nuscenes_expert.py:6-94).  AutoMoE's configuration keeps
it off (`model_config.json`: "use_lidar": false) and the reference feeds zeros when no point cloud is in the batch.
"""
import torch
import torch.nn as nn

from ... import runtime
from ...hip import conv as hconv
from ...hip import ops as hops
from .._nn import Dropout, Linear, RowLinear
from .resnet import Trunk, load_pretrained_


class _PooledTrunk(Trunk):
    """torchvision resnet18 children()[:-1]: the trunk plus AdaptiveAvgPool2d(1) at index 8 (no parameters)."""

    def __init__(self):
        super().__init__()
        self.append(nn.AdaptiveAvgPool2d((1, 1)))

    def forward(self, x):  # NHWC compute dtype -> [B, 512] fp32
        return hops.GapNhwc.apply(Trunk.forward(self, x), runtime.loss_scale())


class NuScenesExpert(nn.Module):
    def __init__(self, image_backbone=None, lidar_backbone=None, fusion: str = "concat", num_queries: int = 100,
                 use_lidar: bool = False, use_tnet: bool = False, bbox_dim: int = 7, pretrained_backbone: bool = True):
        super().__init__()
        if use_lidar or lidar_backbone is not None:
            raise NotImplementedError("NuScenesExpert(use_lidar=True): the PointNet LiDAR branch is not part of this build "
                                      "(AutoMoE runs the expert image-only; see DESIGN.md section 7)")
        if image_backbone is None:
            self.image_backbone = _PooledTrunk()
            load_pretrained_(self.image_backbone, pretrained_backbone)  # the reference fetches ImageNet weights unconditionally
            self.image_projection = Linear(512, 256)
        else:
            self.image_backbone = image_backbone  # caller-supplied: takes the NCHW image, returns [B, 256(,1,1)]
            self.image_projection = nn.Identity()
        self.use_lidar = False
        self.lidar_backbone = None
        self.fusion_type = fusion
        fusion_dim = 256
        self.num_queries = num_queries
        self.bbox_dim = bbox_dim
        self.query_embed = nn.Embedding(num_queries, fusion_dim)
        self.decoder = nn.Sequential(RowLinear(fusion_dim, 256), nn.ReLU(), Dropout(0.3), RowLinear(256, 128), nn.ReLU(),
                                     Dropout(0.3))
        self.class_head = RowLinear(128, 10)
        self.bbox_head = RowLinear(128, self.bbox_dim)

    def forward(self, batch, nhwc_input=None):
        image = batch["image"]
        own_step = nhwc_input is None
        if isinstance(self.image_backbone, _PooledTrunk):
            if own_step:
                runtime.begin_step(image.device)
            xin = nhwc_input if nhwc_input is not None else hops.image_to_nhwc(image, runtime.compute_dtype())
            img_feat = self.image_backbone(xin)
            if own_step:
                hconv.flush_bn_counters()
        else:
            img_feat = self.image_backbone(image)
        img_feat = img_feat.view(img_feat.size(0), -1)
        fused = self.image_projection(img_feat)                       # [B, 256]
        x = fused.unsqueeze(1) + self.query_embed.weight.unsqueeze(0)  # [B, Q, 256]
        x = self.decoder[2](self.decoder[0](x, relu=True))
        x = self.decoder[5](self.decoder[3](x, relu=True))
        return {"class_logits": self.class_head(x), "bbox_preds": self.bbox_head(x)}
