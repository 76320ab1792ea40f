"""Where does conv3x3_c64n64_duo_k's launch go?  1 = no patch DMA, 2 = no MFMA, 4 = no fragment reads, 8 = no global stores."""
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
NAMES = {0: "full", 1: "no DMA", 2: "no MFMA", 4: "no reads", 8: "no stores", 3: "no DMA, no MFMA", 6: "no MFMA, no reads", 9: "no DMA, no stores",
         11: "no DMA/MFMA/stores (reads + epilogue math)", 13: "MFMA only + epilogue math", 14: "DMA only + epilogue math", 7: "stores + epilogue math only"}
B, H, W = int(os.environ.get("B", 32)), 180, 320
s = hc.ConvSpec(64, 64, 3, 1, 1)
x = torch.relu(torch.randn(B, H, W, 64, device=dev)).to(dt)
wp = hc.pack_fwd(torch.randn(64, 64, 3, 3, device=dev) / 24, s, dt)
y = torch.empty(B, H, W, 64, dtype=dt, device=dev)
stats = torch.zeros(16 * 2 * 64, dtype=torch.float64, device=dev)
g = hc.fwd_geom(s, B, H, W, 64, 64, 2)
fl = 2.0 * B * H * W * 64 * 9 * 64
for rep in range(2):
    print(f"B={B} {H}x{W} 64->64 (rep {rep})")
    variants = [(f"duo_abl{m}.so", f"mask {m:2d} {NAMES[m]}") for m in sorted(NAMES)] + [(f, f[:-3]) for f in sorted(os.listdir(HERE)) if f.startswith("duo_") and f.endswith(".so") and "abl" not in f]
    if os.environ.get("ONLY"):
        variants = [v for v in variants if any(k in v[0] for k in os.environ["ONLY"].split(","))]
    for fname, label in variants:
        so = os.path.join(HERE, fname)
        if not os.path.exists(so): continue
        lib = ctypes.CDLL(so); lib.duo_run.restype = ctypes.c_int
        f = lambda: lib.duo_run(ctypes.byref(g), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                ctypes.c_void_p(stats.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert f() == 0
        us = t(f)
        print(f"  {label:52s} {us:7.1f} us  ({fl / us / 1e6:6.0f} 'TF/s', {2 * B * H * W * 64 * 2 / us / 1e6:5.2f} 'TB/s')", flush=True)
