"""1x1 / stride-2 shortcut convolutions (K = 128 / 256: 4 / 8 K-steps): 256x256 ring tile vs the 256x128 tile (AM_TUNE_RING_SHORT_K)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from self_driving_model_amd.hip import conv as hc
L = hc._L(); dev = torch.device("cuda:0"); dt = torch.float16
for B in (32, 16):
    for name, s, IH, IW in (("l3.0 ds 1x1 128->256 s2", hc.ConvSpec(128, 256, 1, 2, 0), 90, 160), ("l4.0 ds 1x1 256->512 s2", hc.ConvSpec(256, 512, 1, 2, 0), 45, 80),
                            ("l2.0 ds 1x1 64->128 s2", hc.ConvSpec(64, 128, 1, 2, 0), 180, 320)):
        x = torch.randn(B, IH, IW, s.cin, device=dev).to(dt)
        w = torch.randn(s.cout, s.cin, 1, 1, device=dev) * 0.05
        wp = hc.pack_fwd(w, s, dt)
        OH, OW = hc.out_size(IH, s), hc.out_size(IW, s)
        g = hc.fwd_geom(s, B, IH, IW, s.cin, s.cout, 2)
        y = torch.empty(B, OH, OW, s.cout, device=dev, dtype=dt)
        stats = torch.zeros(16 * 2 * s.cout, dtype=torch.float64, device=dev)
        res = {}
        ts = {0: [], 8: []}
        for r in range(7):
            for sk in (0, 8):
                L.am_set_tuning(4, sk)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): hc.conv_gemm(g, x, wp, None, False, y, stats)
                e1.record(); torch.cuda.synchronize()
                if r: ts[sk].append(e0.elapsed_time(e1) / 10)
                res[sk] = (L.am_conv_last_variant(), y.float().clone())
        L.am_set_tuning(4, 0)
        d = float((res[0][1] - res[8][1]).abs().max())
        print(f"B={B} {name:26s} 256x256 (kernel {res[0][0]}) {sorted(ts[0])[3]*1e3:6.1f} us | 256x128 (kernel {res[8][0]}) {sorted(ts[8])[3]*1e3:6.1f} us  max diff {d:.1e}")
