import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from self_driving_model_amd import runtime
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_gating_network import GatingTrainStep
dev = torch.device("cuda:0")
model = create_automoe_model(bench.MODEL_CFG, dev); model.freeze_experts(); model.train()
step = GatingTrainStep(model, bench.TRAIN_CFG)
for B in (32, 4):
    batch = synthetic.carla_sequence_batch(B, 720, 1280, 10, dev, seed=0)
    for _ in range(3): step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: enqueue {1e3*(t1-t0)/5:.2f} ms/step, total {1e3*(t2-t0)/5:.2f} ms/step")
import cProfile, pstats
batch = synthetic.carla_sequence_batch(4, 720, 1280, 10, dev, seed=0)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step(batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
