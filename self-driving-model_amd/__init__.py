"""MI355X-native (gfx950) implementation of AutoMoE's data-parallel train-step hot path.

Mirrors the reference's import paths for that path only:
    models.automoe / models.experts / models.gating / models.policy / models.context
    training.hungarian_matcher / training.train_bdd100k_ddp / training.train_gating_network
    inference.run_automoe (load_model / model_infer)
All arithmetic runs in hand-written HIP kernels behind the C ABI in include/automoe_hip.h
(csrc/ -> libautomoe_hip.so); PyTorch-ROCm provides device memory, streams, autograd bookkeeping and
torch.distributed (RCCL).  There is no CPU fallback: the product path raises if the extension is
not built or no HIP device is present.
"""
__version__ = "0.1.0"
