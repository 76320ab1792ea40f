"""First-layer (stem) weight gradient: plain vs fused-BatchNorm-backward form, (round 2: 4 waves / transform at load time 295 / 688 us at B=16 -> 8 waves, transform at store time, taps split over waves 206 / 448 us; ~90 us of either is the final fp32-atomic flush)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc, lib
L = lib.get()
B = int(os.environ.get("B", 16)); dt = torch.float16; dev = torch.device("cuda:0")
COUT, K, PAD = (32, 5, 2) if os.environ.get("POLICY") else (64, 7, 3)
s = hc.ConvSpec(3, COUT, K, 2, PAD, first=True); IH, IW = 720, 1280
OH, OW = hc.out_size(IH, s), hc.out_size(IW, s)
x = torch.randn(B, IH // 2, IW // 2, 16, device=dev).to(dt)
g = hc.fwd_geom(s, B, IH // 2, IW // 2, 16, s.cout, 2, orig_hw=(IH, IW))
dy = torch.randn(B, OH, OW, COUT, device=dev).to(dt); raw = torch.randn_like(dy); y = torch.relu(raw)
mean = torch.zeros(COUT, device=dev); rstd = torch.ones(COUT, device=dev); coef = torch.rand(3 * COUT, device=dev)
dwp = torch.zeros(COUT, g.ntaps * g.krun, dtype=torch.float32, device=dev)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
plain = lambda: L.am_conv_wgrad(ctypes.byref(g), hc.dt_code(dt), hc.ptr(x), hc.ptr(dy), 1.0, hc.ptr(dwp), hc.stream())
bnf = lambda: L.am_conv_wgrad_bn(ctypes.byref(g), hc.dt_code(dt), hc.ptr(x), hc.ptr(dy), hc.ptr(y), hc.ptr(raw), hc.ptr(mean), hc.ptr(rstd), hc.ptr(coef), 1, 1.0, hc.ptr(dwp), hc.stream())
sc, sh = torch.rand(COUT, device=dev) + 0.5, torch.randn(COUT, device=dev)
sg = lambda: L.am_conv_wgrad_bn_sign(ctypes.byref(g), hc.dt_code(dt), hc.ptr(x), hc.ptr(dy), hc.ptr(raw), hc.ptr(mean), hc.ptr(rstd), hc.ptr(coef), hc.ptr(sc), hc.ptr(sh), 1.0, hc.ptr(dwp), hc.stream())
for _ in range(3):
    print(f"B={B}  sign {t(sg):7.1f} us   fused-BN(y) {t(bnf):7.1f} us", flush=True)
print(f"B={B}  plain {t(plain):7.1f} us   fused-BN {t(bnf):7.1f} us", flush=True)
