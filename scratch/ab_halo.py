"""conv_halo_k vs conv_ring_k<256,128> on the layer2 shapes (3x3 / stride 1, 128 -> 128 at 90x160; 64 -> 128 does not exist at stride 1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc, lib
L = lib.get(); dev = torch.device("cuda:0"); dt = torch.float16
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, cin, cout) in [(32, 90, 160, 128, 128), (16, 90, 160, 128, 128), (64, 90, 160, 128, 128), (32, 45, 80, 128, 128)]:
    s = hc.ConvSpec(cin, cout, 3, 1, 1)
    x = torch.randn(B, H, W, cin, device=dev).to(dt)
    w = torch.randn(cout, cin, 3, 3, device=dev) / 34
    wp = hc.pack_fwd(w, s, dt)
    y = torch.empty(B, H, W, cout, dtype=dt, device=dev)
    stats = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=dev)
    g = hc.fwd_geom(s, B, H, W, cin, cout, 2)
    fl = 2.0 * B * H * W * cin * 9 * cout
    res = []
    for mt in (1, 1 << 30):
        old = L.am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, mt)
        us = t(lambda: hc.conv_gemm(g, x, wp, None, False, y, stats))
        k = L.am_conv_last_variant()
        L.am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, old)
        res.append(f"{hc.CONV_KERNEL_NAMES.get(k, k)} {us:7.1f} us {fl / us / 1e6:6.0f} TF/s")
    print(f"B={B} {H}x{W} {cin}->{cout}: " + "   |   ".join(res), flush=True)
# PRE form (BatchNorm + ReLU of the producing layer applied to the staged patch) against the plain form
import ctypes
for (B, H, W, cin, cout) in [(32, 90, 160, 128, 128)]:
    s = hc.ConvSpec(cin, cout, 3, 1, 1)
    x = torch.randn(B, H, W, cin, device=dev).to(dt); w = torch.randn(cout, cin, 3, 3, device=dev) / 34; wp = hc.pack_fwd(w, s, dt)
    y = torch.empty(B, H, W, cout, dtype=dt, device=dev); stats = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev)
    g = hc.fwd_geom(s, B, H, W, cin, cout, 2)
    us_pre = t(lambda: L.am_conv_gemm_prebn(ctypes.byref(g), 1, hc.ptr(x), hc.ptr(sc), hc.ptr(sh), hc.ptr(wp), hc.ptr(y), hc.ptr(stats), hc.stream()))
    us_plain = t(lambda: hc.conv_gemm(g, x, wp, None, False, y, stats))
    print(f"B={B} {H}x{W} {cin}->{cout}: conv_halo_k<PRE> {us_pre:7.1f} us   plain {us_plain:7.1f} us", flush=True)
