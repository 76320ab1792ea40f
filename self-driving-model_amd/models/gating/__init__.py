from .gating_network import ContextEncoder, ExpertOutputProcessor, GatingNetwork

__all__ = ["GatingNetwork", "ContextEncoder", "ExpertOutputProcessor"]
