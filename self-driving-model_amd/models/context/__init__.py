from .context_features import SimpleContextExtractor, create_context_extractor

__all__ = ["SimpleContextExtractor", "create_context_extractor"]
