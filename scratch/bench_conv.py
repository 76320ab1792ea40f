"""Per-layer conv forward micro-benchmark at the bench shapes (B=32, 720p): TFLOP/s per layer shape."""
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import torch
from self_driving_model_amd.hip import conv as hc
from self_driving_model_amd.hip import lib

B = int(os.environ.get("B", 32))
dt = torch.float16
layers = [  # name, spec, IH, IW
    ("stem7x7 3->64 s2", hc.ConvSpec(3, 64, 7, 2, 3, first=True), 720, 1280),
    ("l1 3x3 64->64", hc.ConvSpec(64, 64, 3, 1, 1), 180, 320),
    ("l2.0 3x3 64->128 s2", hc.ConvSpec(64, 128, 3, 2, 1), 180, 320),
    ("l2 3x3 128->128", hc.ConvSpec(128, 128, 3, 1, 1), 90, 160),
    ("l2.0 ds 1x1 64->128 s2", hc.ConvSpec(64, 128, 1, 2, 0), 180, 320),
    ("l3.0 3x3 128->256 s2", hc.ConvSpec(128, 256, 3, 2, 1), 90, 160),
    ("l3.0 ds 1x1 128->256 s2", hc.ConvSpec(128, 256, 1, 2, 0), 90, 160),
    ("l4.0 ds 1x1 256->512 s2", hc.ConvSpec(256, 512, 1, 2, 0), 45, 80),
    ("l3 3x3 256->256", hc.ConvSpec(256, 256, 3, 1, 1), 45, 80),
    ("l4.0 3x3 256->512 s2", hc.ConvSpec(256, 512, 3, 2, 1), 45, 80),
    ("l4 3x3 512->512", hc.ConvSpec(512, 512, 3, 1, 1), 23, 40),
    ("head 3x3 512->256", hc.ConvSpec(512, 256, 3, 1, 1), 23, 40),
    ("pol0 5x5 3->32 s2", hc.ConvSpec(3, 32, 5, 2, 2, first=True), 720, 1280),
    ("pol1 3x3 32->64 s2", hc.ConvSpec(32, 64, 3, 2, 1), 360, 640),
    ("pol2 3x3 64->128 s2", hc.ConvSpec(64, 128, 3, 2, 1), 180, 320),
    ("pol3 3x3 128->256 s2", hc.ConvSpec(128, 256, 3, 2, 1), 90, 160),
]
dev = torch.device("cuda:0")
print(f"B={B}")
tot_ms = tot_fl = 0
for name, s, IH, IW in layers:
    ldi = 16 if s.first else s.cin
    if s.first:
        x = torch.randn(B, IH // 2, IW // 2, 16, device=dev).to(dt)
    else:
        x = torch.randn(B, IH, IW, ldi, device=dev).to(dt)
    w = torch.randn(s.cout, s.cin, s.k, s.k, device=dev) * 0.05
    wp = hc.pack_fwd(w, s, dt)
    OH, OW = hc.out_size(IH, s), hc.out_size(IW, s)
    gIH, gIW = (IH // 2, IW // 2) if s.first else (IH, IW)
    y = torch.empty(B, OH, OW, s.cout, device=dev, dtype=dt)
    stats = torch.zeros(16 * 2 * s.cout, dtype=torch.float64, device=dev)
    g = hc.fwd_geom(s, B, gIH, gIW, ldi, s.cout, 2, orig_hw=(IH, IW))
    for _ in range(3):
        hc.conv_gemm(g, x, wp, None, False, y, stats)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        hc.conv_gemm(g, x, wp, None, False, y, stats)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * OH * OW * s.cin * s.k * s.k * s.cout
    by = (x.numel() + y.numel()) * 2
    tot_ms += ms; tot_fl += fl
    print(f"{name:26s} M={B*OH*OW:8d} K={s.cin*s.k*s.k:5d} N={s.cout:4d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  min-HBM {by/ms/1e6:6.0f} GB/s")
print(f"sum {tot_ms:.2f} ms, {tot_fl/tot_ms/1e9:.1f} TF/s")
