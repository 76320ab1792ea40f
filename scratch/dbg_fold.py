import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import conv as hc
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
runtime.set_compute_dtype(torch.float16)
dev = torch.device("cuda:0")
m = create_automoe_model(bench.MODEL_CFG, dev); m.fuse_expert_pooling = True; m.eval()
cnt = {"fold": 0, "conv": 0, "bn_apply": 0}
orig = hc.conv_gemm
def cg(*a, **k):
    cnt["conv"] += 1; return orig(*a, **k)
hc.conv_gemm = cg
L = hc._L()
oa = L.am_bn_apply
def ba(*a): cnt["bn_apply"] += 1; return oa(*a)
L.am_bn_apply = ba
batch = synthetic.carla_sequence_batch(4, 720, 1280, 10, dev, seed=1)
with torch.no_grad(): m(batch)
print(cnt, sum(1 for mod in m.modules() if getattr(getattr(mod, "_packed", None), "fold", None) is not None))
