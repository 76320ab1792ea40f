"""torch.profiler view of one eager config-4a train step: which aten ops launch the small kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from self_driving_model_amd import runtime
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_gating_network import GatingTrainStep
runtime.set_compute_dtype(torch.float16)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = create_automoe_model(bench.MODEL_CFG, dev); model.freeze_experts(); model.train()
step = GatingTrainStep(model, bench.TRAIN_CFG); step.use_graph = False
batch = synthetic.carla_sequence_batch(8, bench.H, bench.W, 10, dev, seed=0)
for _ in range(3): step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(batch); torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0 or e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[1])
for k, n, t in rows[:45]:
    print(f"{n:5d} {t/1e3:8.3f} ms  {k[:100]}")

# ---- second pass: python call sites of the small-kernel ops ----
import collections
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof2:
    step(batch); torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof2.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::add_", "aten::clone", "aten::cat", "aten::mul", "aten::index_select", "aten::index_copy_", "aten::_to_copy"):
        st = [f for f in (ev.stack or []) if "self-driving-model_amd" in f or "self_driving_model_amd" in f or "bench" in f or "training" in f]
        sites[(ev.name, st[0][-90:] if st else "(autograd engine / no python frame)")] += 1
for (n, s), c in sites.most_common(40):
    print(f"{c:4d} {n:18s} {s}")
