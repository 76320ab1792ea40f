import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests', 'golden'))
import torch, numpy as np
from _seeded import seed_module_, seeded_tensor
from oracle import torch_ref as oref
from self_driving_model_amd import runtime
import self_driving_model_amd.models.experts as hx

def rel(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))

for (H, W, train) in [(128, 160, True), (128, 160, False), (256, 320, True), (384, 640, True)]:
    ref = seed_module_(oref.BDDDetectionExpert(10, False), 36)
    ref.train(train)
    x = seeded_tensor((2, 3, H, W), 37)
    o_r = ref(x)
    pc, pb = seeded_tensor(o_r["class_logits"].shape, 38), seeded_tensor(o_r["bbox_deltas"].shape, 39)
    ((o_r["class_logits"] * pc).sum() + (o_r["bbox_deltas"] * pb).sum()).backward()
    for ls in (1.0, 1024.0):
        hip = hx.BDDDetectionExpert(10, False); hip.load_state_dict(seed_module_(oref.BDDDetectionExpert(10, False), 36).state_dict()); hip.cuda()
        hip.train(train)
        with runtime.precision(torch.float16, ls):
            o = hip(x.cuda())
            ((o["class_logits"] * pc.cuda()).sum() + (o["bbox_deltas"] * pb.cuda()).sum()).backward()
        errs = [(rel(p.grad, q.grad), n) for (n, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters())]
        e = np.array([a for a, _ in errs])
        print(f"H={H} W={W} train={train} ls={ls}: fwd {rel(o['class_logits'], o_r['class_logits']):.2e}  grad rel err median {np.median(e):.3e} max {e.max():.3e} ({errs[int(e.argmax())][1]}) head.2.w {errs[-2][0]:.2e} conv1 {errs[0][0]:.2e}")
