from .bdd_detection_expert import BDDDetectionExpert
from .bdd_drivable_expert import BDDDrivableExpert
from .bdd_segmentation_expert import BDDSegmentationExpert
from .nuscenes_expert import NuScenesExpert
from .expert_extractors import (DetectionExpertExtractor, DrivableExpertExtractor, NuScenesExpertExtractor,
                                SegmentationExpertExtractor, create_expert_extractors)

__all__ = ["BDDDetectionExpert", "BDDDrivableExpert", "BDDSegmentationExpert", "NuScenesExpert", "DetectionExpertExtractor",
           "SegmentationExpertExtractor", "DrivableExpertExtractor", "NuScenesExpertExtractor", "create_expert_extractors"]
