"""BASELINE configs[3] variant 4b (experts unfrozen), B=32: img/s of the hipGraph step (for rocprofv3: python3 scratch/run_4b.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from self_driving_model_amd import runtime
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_gating_network import GatingTrainStep
runtime.set_compute_dtype(torch.float16)
dev = torch.device("cuda:0")
m = create_automoe_model(bench.MODEL_CFG, dev)
m.fuse_expert_pooling = True
m.unfreeze_experts()
batch = synthetic.carla_sequence_batch(32, bench.H, bench.W, 10, dev, seed=0)
step = GatingTrainStep(m, bench.TRAIN_CFG)
n = int(os.environ.get("STEPS", 5))
dt = bench.timed_steps(lambda: step(batch), n, 4, False)
print("4b img/s", round(32 * n / dt, 2), flush=True)
