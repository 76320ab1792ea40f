"""Which C-ABI calls the MoE tail (extractor / gating MLPs, gate combine, dropout, LayerNorm) makes in one eager 4a step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import lib as hlib
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_gating_network import GatingTrainStep
runtime.set_compute_dtype(torch.float16)
dev = torch.device("cuda:0")
m = create_automoe_model(bench.MODEL_CFG, dev); m.freeze_experts(); m.train()
step = GatingTrainStep(m, bench.TRAIN_CFG, use_graph=False)
step.prefetch_experts = False
batch = synthetic.carla_sequence_batch(8, bench.H, bench.W, 10, dev, seed=0)
for _ in range(2): step(batch)
hlib.CALL_COUNTS = {}
step(batch); torch.cuda.synchronize()
calls, hlib.CALL_COUNTS = hlib.CALL_COUNTS, None
tail = ("am_linear", "am_layernorm", "am_moe_tail", "am_gate", "am_dropout")
print("tail calls:", sum(v for k, v in calls.items() if k.startswith(tail)))
for k, v in sorted(calls.items(), key=lambda kv: -kv[1]):
    print(f"{v:4d} {k}{'   <- tail' if k.startswith(tail) else ''}")
print("all:", sum(calls.values()))
