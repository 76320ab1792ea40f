"""Where does conv_band16_k's K-step go?  The product kernel rebuilt with parts of the K-loop removed (results are garbage, the
timing is the point): 1 = no LDS-DMA, 2 = no barrier, 4 = no fragment reads, 8 = no MFMAs.  Run on the GPU box:
    bash scratch/ablate_band16/build.sh && python scratch/ablate_band16/run.py"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from self_driving_model_amd.hip import conv as hc
dev = torch.device("cuda:0"); dt = torch.float16
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
NAMES = {0: "full", 1: "no DMA", 2: "no barrier", 4: "no fragment reads", 8: "no MFMA", 3: "no DMA, no barrier", 5: "no DMA, no reads", 12: "no reads, no MFMA (DMA + barrier)",
         7: "MFMA only", 6: "no barrier, no reads", 9: "no DMA, no MFMA (reads + barrier)", 14: "DMA only", 10: "no barrier, no MFMA"}
for (B, H, W, cin, cout) in [(32, 45, 80, 256, 256), (32, 23, 40, 512, 512)]:
    s = hc.ConvSpec(cin, cout, 3, 1, 1)
    x = torch.relu(torch.randn(B, H, W, cin, device=dev)).to(dt)
    wp = hc.pack_fwd(torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5), s, dt)
    y = torch.empty(B, H, W, cout, dtype=dt, device=dev)
    stats = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=dev)
    g = hc.fwd_geom(s, B, H, W, cin, cout, 2)
    fl = 2.0 * B * H * W * cin * 9 * cout
    print(f"B={B} {H}x{W} {cin}->{cout}")
    variants = [(f"band16_abl{m}.so", f"mask {m:2d} {NAMES[m]}") for m in sorted(NAMES)]
    variants += [(f, f[:-3]) for f in sorted(os.listdir(HERE)) if f.startswith("band16_") and f.endswith(".so") and not f.startswith("band16_abl")]
    if os.environ.get("ONLY"):
        variants = [v for v in variants if os.environ["ONLY"] in v[0]]
    for fname, label in variants:
        so = os.path.join(HERE, fname)
        if not os.path.exists(so):
            continue
        lib = ctypes.CDLL(so)
        lib.band16_run.restype = ctypes.c_int
        f = lambda: lib.band16_run(ctypes.byref(g), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                   ctypes.c_void_p(stats.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert f() == 0
        us = t(f)
        print(f"  {label:44s} {us:7.1f} us   ({fl / us / 1e6:6.0f} 'TF/s')", flush=True)
