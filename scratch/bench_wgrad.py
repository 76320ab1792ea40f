"""Per-layer conv weight-gradient micro-benchmark at the bench shapes (B=32, 720p)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc
B = int(os.environ.get("B", 32)); dt = torch.float16; dev = torch.device("cuda:0")
layers = [("l1 3x3 64->64", hc.ConvSpec(64, 64, 3, 1, 1), 180, 320), ("l2.0 3x3 64->128 s2", hc.ConvSpec(64, 128, 3, 2, 1), 180, 320),
          ("l2 3x3 128->128", hc.ConvSpec(128, 128, 3, 1, 1), 90, 160), ("l3 3x3 256->256", hc.ConvSpec(256, 256, 3, 1, 1), 45, 80),
          ("l4 3x3 512->512", hc.ConvSpec(512, 512, 3, 1, 1), 23, 40), ("pol1 3x3 32->64 s2", hc.ConvSpec(32, 64, 3, 2, 1), 360, 640),
          ("pol2 3x3 64->128 s2", hc.ConvSpec(64, 128, 3, 2, 1), 180, 320), ("pol3 3x3 128->256 s2", hc.ConvSpec(128, 256, 3, 2, 1), 90, 160)]
layers = [("pol0 5x5 3->32 s2 (s2d)", hc.ConvSpec(3, 32, 5, 2, 2, first=True), 720, 1280),
          ("stem 7x7 3->64 s2 (s2d)", hc.ConvSpec(3, 64, 7, 2, 3, first=True), 720, 1280)] + layers
tot = 0
for name, s, IH, IW in layers:
    OH, OW = hc.out_size(IH, s), hc.out_size(IW, s)
    if s.first:
        x = torch.randn(B, IH // 2, IW // 2, 16, device=dev).to(dt)
        g = hc.fwd_geom(s, B, IH // 2, IW // 2, 16, s.cout, 2, orig_hw=(IH, IW))
    else:
        x = torch.randn(B, IH, IW, s.cin, device=dev).to(dt)
        g = hc.fwd_geom(s, B, IH, IW, s.cin, s.cout, 2)
    dy = torch.randn(B, OH, OW, s.cout, device=dev).to(dt)
    dwp = torch.zeros(s.cout, g.ntaps * g.krun, dtype=torch.float32, device=dev)
    for _ in range(3): hc.conv_wgrad(g, x, dy, 1.0, dwp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hc.conv_wgrad(g, x, dy, 1.0, dwp)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * OH * OW * s.cin * s.k * s.k * s.cout
    tot += ms
    print(f"{name:24s} M={B*OH*OW:8d} K={s.cin*s.k*s.k:5d} N={s.cout:4d} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s  in+dy {((x.numel()+dy.numel())*2)/ms/1e6:6.0f} GB/s")
print(f"sum {tot:.2f} ms")
