"""Experiment: run the main step graph (B) and the expert-prefetch graph (A) on CU-masked streams (disjoint CU sets)."""
import sys, os, time, ctypes
sys.path.insert(0, os.getcwd())
import torch
import bench
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import lib as hlib
from self_driving_model_amd.models.automoe import create_automoe_model
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_gating_network import GatingTrainStep
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
hlib.get()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)
mode = os.environ.get("MASK", "none")
runtime.set_compute_dtype(torch.float16)
model = create_automoe_model(bench.MODEL_CFG, dev); model.freeze_experts(); model.train()
step = GatingTrainStep(model, bench.TRAIN_CFG)
batch = synthetic.carla_sequence_batch(32, bench.H, bench.W, 10, dev, seed=0)
for i in range(4):
    step(step.input_buffers or batch, next_batch=True)
torch.cuda.synchronize()
sb = sa = None
if mode != "none":
    wb = int(mode, 16)                      # per-word mask of graph B's stream, e.g. 11111111 (every 4th CU)
    sb = masked_stream([wb] * 8)
    sa = masked_stream([(~wb) & 0xFFFFFFFF] * 8)
    step._expert_stream = sa
def run():
    if sb is None:
        step(step.input_buffers or batch, next_batch=True)
    else:
        cur = torch.cuda.current_stream()
        sb.wait_stream(cur)
        with torch.cuda.stream(sb):
            step(step.input_buffers or batch, next_batch=True)
        cur.wait_stream(sb)
for _ in range(5): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
print("MASK=%s  %.2f ms/step  %.1f img/s" % (mode, dt * 1e3, 32 / dt))
