import sys, os
sys.path.insert(0, os.getcwd())
import torch
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import ops as hops
dev = torch.device("cuda:0")
img = torch.randn(int(os.environ.get("B", 32)), 3, 720, 1280, device=dev)
with runtime.precision(torch.float16):
    for _ in range(3): x = hops.image_to_s2d(img, torch.float16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): x = hops.image_to_s2d(img, torch.float16)
    e1.record(); torch.cuda.synchronize()
    print("image_s2d %.1f us, %.2f TB/s" % (e0.elapsed_time(e1) / 20 * 1e3, (img.numel() * 4 + x.numel() * 2) / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12))
    # reference layout check
    B, C, H, W = img.shape
    ref = torch.zeros(B, H // 2, W // 2, 16, device=dev, dtype=torch.float16)
    for py in range(2):
        for px in range(2):
            ref[..., (py * 2 + px) * 3:(py * 2 + px) * 3 + 3] = img[:, :, py::2, px::2].permute(0, 2, 3, 1).half()
    print("exact:", torch.equal(ref, x))
