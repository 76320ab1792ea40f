"""One instrumented eager AutoMoE step, experts trainable as the model is built (streams serialised): every conv launch with its kernel, algorithmic GFLOP, microseconds, TFLOP/s."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import conv as hconv
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_gating_network import GatingTrainStep
runtime.set_compute_dtype(torch.float16)
dev = torch.device("cuda:0")
m = create_automoe_model(bench.MODEL_CFG, dev); m.fuse_expert_pooling = True
if os.environ.get("FROZEN", "1") == "1":
    m.freeze_experts()
m.train()
m.parallel_experts = m.overlap_policy_backbone = False
batch = synthetic.carla_sequence_batch(32, bench.H, bench.W, 10, dev, seed=0)
step = GatingTrainStep(m, bench.TRAIN_CFG, use_graph=False)
step.prefetch_experts = False
for _ in range(2): step(batch)
hconv.TIMER = hconv.KernelTimer()
step(batch)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, flops, e0, e1, kernel, nbytes in hconv.TIMER.records:
    us = e0.elapsed_time(e1) * 1e3
    key = (kind, kernel, round(flops / 1e9, 1), round(nbytes / 1e6, 1))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += us
for (kind, kernel, gf, mb), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{kind:11s} {kernel:34s} {gf:8.1f} GF x{n:2d}  {us / n:8.1f} us each  {gf / (us / n) * 1e3 if us else 0:7.1f} TF/s  {mb:7.1f} MB {mb / (us / n) * 1e-3 if us else 0:5.2f} TB/s  total {us / 1e3:6.3f} ms")
