"""Count where the small torch ops of one eager config-4a step come from (python call sites)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from self_driving_model_amd import runtime
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_gating_network import GatingTrainStep
runtime.set_compute_dtype(torch.float16)
dev = torch.device("cuda", 0)
model = create_automoe_model(bench.MODEL_CFG, dev); model.freeze_experts(); model.train()
step = GatingTrainStep(model, bench.TRAIN_CFG); step.use_graph = False
batch = synthetic.carla_sequence_batch(4, bench.H, bench.W, 10, dev, seed=0)
for _ in range(3): step(batch)
sites = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "fill_", "zero_", "add_", "add.", "cat", "mul", "zeros", "index", "stack", "sum", "mean", "div", "sub", "neg", "abs", "where", "_to_copy")):
            fr = [f for f in traceback.extract_stack() if ("self-driving-model_amd" in f.filename or f.filename.endswith("bench.py")) ]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(engine)"
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    step(batch)
torch.cuda.synchronize()
for (n, w), c in sites.most_common(70):
    print(f"{c:4d} {n:34s} {w}")
