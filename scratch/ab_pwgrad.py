"""layer1 (64 -> 64, 3x3 / s1) weight gradient: conv_patch_wgrad_k vs the generic conv_wgrad_k, workspace and atomic forms."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc, lib
L = lib.get()
B = int(os.environ.get("B", 16)); C = int(os.environ.get("C", 64)); H, W = (180, 320) if C == 64 else (90, 160)
dt = torch.float16; dev = torch.device("cuda:0")
s = hc.ConvSpec(C, C, 3, 1, 1)
x = torch.randn(B, H, W, C, device=dev).to(dt); dy = torch.randn(B, H, W, C, device=dev).to(dt)
g = hc.fwd_geom(s, B, H, W, C, C, 2)
wparam = torch.nn.Parameter(torch.zeros(C, C, 3, 3, device=dev))
dwp = torch.zeros(C, 9 * C, dtype=torch.float32, device=dev)
def t(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * B * H * W * C * 9 * C
for name, mt in (("patch", 1), ("generic", 1 << 30), ("patch", 1), ("generic", 1 << 30)):
    L.am_set_tuning(6, mt)
    ws = t(lambda: hc.conv_wgrad_oihw(g, x, dy, 1.0, wparam, s))
    at = t(lambda: hc.conv_wgrad(g, x, dy, 1.0, dwp))
    print(f"C={C} B={B} {name:8s} workspace form {ws:7.1f} us ({fl / ws * 1e-6:6.0f} TFLOP/s)   atomic form {at:7.1f} us", flush=True)
