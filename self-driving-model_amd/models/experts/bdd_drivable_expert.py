"""BDDDrivableExpert -- drop-in for models/experts/bdd_drivable_expert.py:5-23."""
from .bdd_segmentation_expert import _DenseExpert


class BDDDrivableExpert(_DenseExpert):
    def __init__(self, num_classes=3, pretrained_backbone=True):
        super().__init__(num_classes, pretrained_backbone)
