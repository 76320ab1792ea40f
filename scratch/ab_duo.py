"""conv3x3_c64n64_duo_k: 32x32x16 (AUTOMOE_TUNE_DUO_MFMA16=0) vs 16x16x32 MFMA form on the layer1 shape; held clock via the wall time of the same work."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc, lib
L = lib.get(); dev = torch.device("cuda:0"); dt = torch.float16
B, H, W = int(os.environ.get("B", 32)), 180, 320
s = hc.ConvSpec(64, 64, 3, 1, 1)
x = torch.randn(B, H, W, 64, device=dev).clamp_min(0).to(dt)
w = torch.randn(64, 64, 3, 3, device=dev) / 24
wp = hc.pack_fwd(w, s, dt)
g = hc.fwd_geom(s, B, H, W, 64, 64, 2)
y = torch.empty(B, H, W, 64, device=dev, dtype=dt)
stats = torch.zeros(16 * 2 * 64, dtype=torch.float64, device=dev)
def t(n=30):
    f = lambda: hc.conv_gemm(g, x, wp, None, False, y, stats)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for m in (0, 1, 0, 1):
    L.am_set_tuning(8, m)
    us = t()
    print(f"B={B} MFMA16={m}: {us:7.1f} us  ({2.0 * B * H * W * 64 * 576 / us * 1e-6:6.0f} TFLOP/s)  variant {L.am_conv_last_variant()}", flush=True)
