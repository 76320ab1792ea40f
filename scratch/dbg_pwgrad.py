import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from self_driving_model_amd.hip import conv as hc, lib
L = lib.get(); dev = torch.device("cuda:0")
L.am_set_tuning(6, 1)
SYNC = os.environ.get("SYNC") == "1"
for (B, H, W) in [(3, 45, 70), (5, 19, 33), (2, 64, 96)]:
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, 64, H, W, generator=g).half().float()
    dyr = (torch.randn(B, 64, H, W, generator=g) * 0.5).half().float()
    spec = hc.ConvSpec(64, 64, 3, 1, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(dev); dyd = dyr.permute(0, 2, 3, 1).contiguous().half().to(dev)
    geom = hc.fwd_geom(spec, B, H, W, 64, 64, 2)
    wparam = torch.nn.Parameter(torch.zeros(64, 64, 3, 3, device=dev))
    outs = []
    for i in range(5):
        outs.append(hc.conv_wgrad_oihw(geom, xd, dyd, 0.5, wparam, spec).clone())
        if SYNC: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print((B, H, W), "sync" if SYNC else "nosync", "ndiff vs run0:", [int((o != outs[0]).sum()) for o in outs], flush=True)
