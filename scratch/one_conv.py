import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc
B = 32; dt = torch.float16; dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "l2"
cfgs = {"l2": (hc.ConvSpec(128, 128, 3, 1, 1), 90, 160), "l3": (hc.ConvSpec(256, 256, 3, 1, 1), 45, 80), "l1": (hc.ConvSpec(64, 64, 3, 1, 1), 180, 320), "l4": (hc.ConvSpec(512, 512, 3, 1, 1), 23, 40)}
s, IH, IW = cfgs[which]
x = torch.randn(B, IH, IW, s.cin, device=dev).to(dt)
w = torch.randn(s.cout, s.cin, s.k, s.k, device=dev) * 0.05
wp = hc.pack_fwd(w, s, dt)
y = torch.empty(B, IH, IW, s.cout, device=dev, dtype=dt)
stats = torch.zeros(16 * 2 * s.cout, dtype=torch.float64, device=dev)
g = hc.fwd_geom(s, B, IH, IW, s.cin, s.cout, 2)
for _ in range(5):
    hc.conv_gemm(g, x, wp, None, False, y, stats)
torch.cuda.synchronize()
