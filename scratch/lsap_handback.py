"""Why does the split assignment solver hand images of the cfg3 step back to the general kernel?  Runs a few cfg3 steps, keeps each
step's cost matrices (matcher.keep_last) and replays them through the CPU model of the split solver (tests/test_lsap_split_model_cpu.py)."""
import sys, os, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, bench
from self_driving_model_amd import runtime
from self_driving_model_amd.models.experts import BDDDetectionExpert
from self_driving_model_amd.training import synthetic
from self_driving_model_amd.training.train_bdd100k_ddp import BDDTrainer
from test_lsap_split_model_cpu import split_solve
runtime.set_compute_dtype(torch.float16)
dev = torch.device("cuda:0")
m = BDDDetectionExpert(10, pretrained_backbone=False).to(dev).train()
b = synthetic.bdd_detection_batch(8, bench.H, bench.W, 10, 32, dev, seed=0)
loader = synthetic.SyntheticLoader(b, 20)
tr = BDDTrainer("detection", m, loader, loader, dev, {"learning_rate": 2e-4, "weight_decay": 1e-5, "epochs": 1, "run_name": "x"})
tr.matcher.keep_last = True
n_tgt = (b["labels"] != -1).sum(dim=1).cpu().tolist()
print("targets per image", n_tgt)
for step in range(12):
    tr.train_step(tr.input_buffers or b)
    torch.cuda.synchronize()
    cost = tr.matcher.last_cost.float().cpu().numpy()  # [B, Nmax, Q]
    why = collections.Counter()
    for i, n in enumerate(n_tgt):
        C = cost[i, :n, :]  # rows = targets (short side), columns = queries: the solver's wide orientation
        got, reason = split_solve(np.ascontiguousarray(C))
        why[reason or "solved"] += 1
        if reason and step in (0, 11):
            r0 = np.sort(C[0])[:6]
            print(f"   step {step} image {i}: {reason}; row 0 cheapest six: {r0}; distinct values in row 0: {len(np.unique(C[0]))} of {C.shape[1]}")
    print("step", step, dict(why), flush=True)
