from .bdd_detection_expert import BDDDetectionExpert
from .bdd_drivable_expert import BDDDrivableExpert
from .bdd_segmentation_expert import BDDSegmentationExpert
from .expert_extractors import (DetectionExpertExtractor, DrivableExpertExtractor, SegmentationExpertExtractor,
                                create_expert_extractors)

__all__ = ["BDDDetectionExpert", "BDDDrivableExpert", "BDDSegmentationExpert", "DetectionExpertExtractor",
           "SegmentationExpertExtractor", "DrivableExpertExtractor", "create_expert_extractors"]
