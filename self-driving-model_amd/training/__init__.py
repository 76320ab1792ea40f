from .hungarian_matcher import HungarianMatcher

__all__ = ["HungarianMatcher"]
