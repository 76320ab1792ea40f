"""Per-step wall time of the bench step right after capture (how many replays until the steady state)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from self_driving_model_amd import runtime
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_gating_network import GatingTrainStep
dev = torch.device("cuda:0")
runtime.set_compute_dtype(torch.float16)
model = create_automoe_model(bench.MODEL_CFG, dev); model.freeze_experts(); model.train()
step = GatingTrainStep(model, bench.TRAIN_CFG)
batch = synthetic.carla_sequence_batch(32, bench.H, bench.W, 10, dev, seed=0)
ts = []
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step(step.input_buffers or batch, next_batch=True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms (sync after each):", " ".join("%.1f" % t for t in ts))
t0 = time.perf_counter()
for i in range(20):
    step(step.input_buffers or batch, next_batch=True)
torch.cuda.synchronize()
print("pipelined 20 steps: %.2f ms/step" % ((time.perf_counter() - t0) * 1e3 / 20))
