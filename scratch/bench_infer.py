import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from self_driving_model_amd import runtime
from self_driving_model_amd.models.automoe import create_automoe_model
runtime.set_compute_dtype(torch.float16)
m = create_automoe_model(bench.MODEL_CFG, torch.device("cuda:0"))
m.fuse_expert_pooling = True
print(os.environ.get("AM_FOLD_EVAL_BN", "1"), bench.bench_inference(m, 64, 20, 3))
