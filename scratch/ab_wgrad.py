"""Weight gradient: atomic form (am_conv_wgrad + zero fill + re-layout, as autograd used it) vs the workspace form
(am_conv_wgrad_ws: slabs + summing pass, OIHW out) per layer shape; interleaved rounds, median."""
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import torch
from self_driving_model_amd.hip import conv as hc
B = int(os.environ.get("B", 32)); dt = torch.float16; dev = torch.device("cuda:0")
layers = [("l1 3x3 64->64", hc.ConvSpec(64, 64, 3, 1, 1), 180, 320), ("l2.0 3x3 64->128 s2", hc.ConvSpec(64, 128, 3, 2, 1), 180, 320),
          ("l2 3x3 128->128", hc.ConvSpec(128, 128, 3, 1, 1), 90, 160), ("l3.0 3x3 128->256 s2", hc.ConvSpec(128, 256, 3, 2, 1), 90, 160),
          ("l3 3x3 256->256", hc.ConvSpec(256, 256, 3, 1, 1), 45, 80), ("l3.0 ds 1x1 128->256", hc.ConvSpec(128, 256, 1, 2, 0), 90, 160),
          ("l4 3x3 512->512", hc.ConvSpec(512, 512, 3, 1, 1), 23, 40), ("head 3x3 512->256", hc.ConvSpec(512, 256, 3, 1, 1), 23, 40),
          ("pol1 3x3 32->64 s2", hc.ConvSpec(32, 64, 3, 2, 1), 360, 640), ("pol2 3x3 64->128 s2", hc.ConvSpec(64, 128, 3, 2, 1), 180, 320),
          ("pol3 3x3 128->256 s2", hc.ConvSpec(128, 256, 3, 2, 1), 90, 160)]
L = hc._L()
tot = [0.0, 0.0]
for name, s, IH, IW in layers:
    OH, OW = hc.out_size(IH, s), hc.out_size(IW, s)
    x = torch.randn(B, IH, IW, s.cin, device=dev).to(dt)
    g = hc.fwd_geom(s, B, IH, IW, s.cin, s.cout, 2)
    dy = torch.randn(B, OH, OW, s.cout, device=dev).to(dt)
    wparam = torch.nn.Parameter(torch.zeros(s.cout, s.cin, s.k, s.k, device=dev))
    nb = ctypes.c_longlong(0)
    L.am_conv_wgrad_workspace_bytes(ctypes.byref(g), 1, ctypes.byref(nb))

    def atomic():
        dwp = torch.zeros(s.cout, g.ntaps * g.krun, dtype=torch.float32, device=dev)
        hc.conv_wgrad(g, x, dy, 1.0, dwp)
        return hc.unpack_wgrad(dwp, s, dt)

    def ws():
        return hc.conv_wgrad_oihw(g, x, dy, 1.0, wparam, s)

    a, b = atomic(), ws()
    k1 = L.am_conv_last_variant()
    err = float((a - b).norm() / a.norm())
    times = [[], []]
    for r in range(6):
        for i, fn in enumerate((atomic, ws)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6): fn()
            e1.record(); torch.cuda.synchronize()
            if r: times[i].append(e0.elapsed_time(e1) / 6)
    med = [sorted(t)[len(t) // 2] for t in times]
    tot[0] += med[0]; tot[1] += med[1]
    fl = 2.0 * B * OH * OW * s.cin * s.k * s.k * s.cout
    print(f"{name:24s} M={B*OH*OW:8d} K={s.cin*s.k*s.k:5d} N={s.cout:4d} kernel {k1:2d} slabs {nb.value // (s.cout * g.ntaps * g.krun * 4):4d} ({nb.value / 1e6:6.1f} MB) "
          f"atomic+fill+relayout {med[0]*1e3:7.1f} us | workspace {med[1]*1e3:7.1f} us ({fl/med[1]/1e9:5.0f} TF)  rel diff {err:.1e}", flush=True)
print(f"sum atomic {tot[0]:.3f} ms, workspace {tot[1]:.3f} ms")
