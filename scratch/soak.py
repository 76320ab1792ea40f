"""Soak: 200 steps of the unfrozen AutoMoE step (4b, B=32) and 300 of the drivable-expert step (cfg2, B=16) on one fixed synthetic
batch each: the loss must stay finite and fall (the model overfits the batch), the loss scale must not collapse."""
import sys, os
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
first = last = None
for i in range(200):
    out = step(batch)
    if i % 25 == 0 or i == 199:
        torch.cuda.synchronize()
        key = next(k for k in ("total_loss", "total", "loss") if k in out) if isinstance(out, dict) else None
        loss = float(out[key] if key else out)
        print(f"4b step {i:4d} loss {loss:.5f} loss_scale {runtime.loss_scale():.0f}", flush=True)
        first = loss if first is None else first; last = loss
        assert loss == loss and abs(loss) < 1e6, "loss is not finite"
assert last < first, (first, last)
print("4b soak ok", first, "->", last, flush=True)

# ---- cfg2 and cfg3: expert trainers ----
from self_driving_model_amd.models.experts import BDDDrivableExpert, BDDDetectionExpert
from self_driving_model_amd.training.train_bdd100k_ddp import BDDTrainer
del step, m, batch
torch.cuda.empty_cache()
for name, model, b in (("cfg2", BDDDrivableExpert(3, pretrained_backbone=False), synthetic.bdd_drivable_batch(16, bench.H, bench.W, 3, dev, seed=0)),
                       ("cfg3", BDDDetectionExpert(10, pretrained_backbone=False), synthetic.bdd_detection_batch(8, bench.H, bench.W, 10, 32, dev, seed=0))):
    model = model.to(dev).train()
    loader = synthetic.SyntheticLoader(b, 300)
    tr = BDDTrainer("drivable" if name == "cfg2" else "detection", model, loader, loader, dev, {"learning_rate": 2e-4, "weight_decay": 1e-5, "epochs": 1, "run_name": "soak"})
    first = last = None
    for i in range(300):
        out = tr.train_step(tr.input_buffers or b)
        if i % 50 == 0 or i == 299:
            torch.cuda.synchronize()
            loss = float(out["loss"] if isinstance(out, dict) else out)
            print(f"{name} step {i:4d} loss {loss:.5f} loss_scale {runtime.loss_scale():.0f}", flush=True)
            first = loss if first is None else first; last = loss
            assert loss == loss and abs(loss) < 1e6, "loss is not finite"
    assert last < first, (name, first, last)
    print(name, "soak ok", first, "->", last, flush=True)
    del tr, model
    torch.cuda.empty_cache()
