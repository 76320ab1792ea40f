"""Fused upsample + cross-entropy kernel at the cfg2 / segmentation sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc
L = hc._L(); dev = torch.device("cuda:0"); st = hc.stream(); p = lambda a: a.data_ptr()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, C, ld) in ((16, 3, 32), (4, 19, 32), (16, 19, 32)):
    h, w, H, W = 23, 40, 720, 1280
    low = torch.randn(B, h, w, ld, device=dev).half(); tgt = torch.randint(0, C, (B, H, W), device=dev)
    acc = torch.zeros(2, dtype=torch.float64, device=dev); G = torch.empty(B, h, w, C, device=dev)
    print(f"B={B} C={C}: upsample_ce2d fwd {t(lambda: L.am_upsample_ce2d_fwd(1, p(low), ld, p(tgt), B, C, h, w, H, W, 255, p(acc), p(G), st)):.1f} us  (labels {tgt.numel() * 8 / 1e6:.0f} MB)", flush=True)
