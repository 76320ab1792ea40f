"""conv_band16_k vs conv_ring16_k<256,256> / conv_ring_k<256,128> on the layer3 / layer4 / head shapes (3x3 / stride 1, N = 256 / 512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc, lib
L = lib.get(); dev = torch.device("cuda:0"); dt = torch.float16
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, cin, cout) in [(32, 45, 80, 256, 256), (32, 23, 40, 512, 512), (32, 23, 40, 512, 256), (16, 45, 80, 256, 256), (16, 23, 40, 512, 512),
                             (8, 45, 80, 256, 256), (8, 23, 40, 512, 512), (64, 45, 80, 256, 256), (64, 23, 40, 512, 512)]:
    s = hc.ConvSpec(cin, cout, 3, 1, 1)
    x = torch.relu(torch.randn(B, H, W, cin, device=dev)).to(dt)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    wp = hc.pack_fwd(w, s, dt)
    y = torch.empty(B, H, W, cout, dtype=dt, device=dev)
    stats = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=dev)
    g = hc.fwd_geom(s, B, H, W, cin, cout, 2)
    fl = 2.0 * B * H * W * cin * 9 * cout
    res = []
    for mt in (1, 1 << 30):
        old = L.am_set_tuning(lib.AM_TUNE_BAND_MIN_TILES, mt)
        us = t(lambda: hc.conv_gemm(g, x, wp, None, False, y, stats))
        k = L.am_conv_last_variant()
        L.am_set_tuning(lib.AM_TUNE_BAND_MIN_TILES, old)
        res.append(f"{hc.CONV_KERNEL_NAMES.get(k, k)} {us:7.1f} us {fl / us / 1e6:6.0f} TF/s")
    print(f"B={B} {H}x{W} {cin}->{cout}: " + "   |   ".join(res), flush=True)
