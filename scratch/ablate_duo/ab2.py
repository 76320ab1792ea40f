"""conv3x3_c64n64_duo_k A/B between builds (scratch/ablate_duo/ab_*.so from wrap2.hip): outputs and BatchNorm statistics must be
bit-identical to ab_head.so (same products, same order), plain and PRE form, on an interior + edge tile shape; then interleaved timing."""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from self_driving_model_amd.hip import conv as hc
dev = torch.device("cuda:0"); dt = torch.float16
libs = {f[:-3]: ctypes.CDLL(os.path.join(HERE, f)) for f in sorted(os.listdir(HERE)) if f.startswith("ab_") and f.endswith(".so")}
s = hc.ConvSpec(64, 64, 3, 1, 1)
P = ctypes.c_void_p
def run(lib, pre, g, x, wp, y, stats, sc, sh):
    st = P(torch.cuda.current_stream().cuda_stream)
    if pre:
        return lib.duo_run_pre(ctypes.byref(g), P(x.data_ptr()), P(sc.data_ptr()), P(sh.data_ptr()), P(wp.data_ptr()), P(y.data_ptr()), P(stats.data_ptr()), st)
    return lib.duo_run(ctypes.byref(g), P(x.data_ptr()), P(wp.data_ptr()), P(y.data_ptr()), P(stats.data_ptr()), st)
torch.manual_seed(0)
for (B, H, W) in ((36, 37, 53), (2, 180, 320), (40, 41, 40), (5, 121, 111)):
    x = torch.randn(B, H, W, 64, device=dev).to(dt)
    wp = hc.pack_fwd(torch.randn(64, 64, 3, 3, device=dev) / 24, s, dt)
    sc = (torch.rand(64, device=dev) + 0.5) * torch.where(torch.rand(64, device=dev) < 0.2, -1.0, 1.0)
    sh = torch.randn(64, device=dev) * 0.3
    g = hc.fwd_geom(s, B, H, W, 64, 64, 2)
    for pre in (False, True):
        outs = {}
        for name, lib in libs.items():
            y = torch.full((B, H, W, 64), 7.0, dtype=dt, device=dev)
            stats = torch.zeros(16 * 2 * 64, dtype=torch.float64, device=dev)
            assert run(lib, pre, g, x, wp, y, stats, sc, sh) == 0
            torch.cuda.synchronize()
            outs[name] = (y.clone(), stats.view(16, 2, 64).sum(0).clone())
        ref = outs["ab_head"]
        xin = torch.relu(x.float() * sc + sh).to(dt).float() if pre else x.float()
        yr = torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2), hc.unpack_fwd(wp, s).float() if hasattr(hc, "unpack_fwd") else None, padding=1).permute(0, 2, 3, 1) if hasattr(hc, "unpack_fwd") else None
        for name, (y, st) in outs.items():
            same = bool((y == ref[0]).all()); sd = float((st - ref[1]).abs().max() / ref[1].abs().max())
            print(f"B={B} {H}x{W} pre={pre} {name:12s} y bit-equal to head: {same}; stats rel diff {sd:.1e}", flush=True)
            err = float((y.float() - ref[0].float()).abs().max() / ref[0].float().abs().max())
            assert ("_t" in name) or (err < 2e-3 and sd < 1e-4), (name, err, sd)
            if not same:
                print(f"      max |y - y_head| / max |y_head| = {err:.2e}")
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H, W = 32, 180, 320
x = torch.relu(torch.randn(B, H, W, 64, device=dev)).to(dt)
wp = hc.pack_fwd(torch.randn(64, 64, 3, 3, device=dev) / 24, s, dt)
sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev) * 0.3
y = torch.empty(B, H, W, 64, dtype=dt, device=dev)
stats = torch.zeros(16 * 2 * 64, dtype=torch.float64, device=dev)
g = hc.fwd_geom(s, B, H, W, 64, 64, 2)
fl = 2.0 * B * H * W * 64 * 9 * 64
for rep in range(3):
    for pre in (False, True):
        for name, lib in libs.items():
            us = t(lambda: run(lib, pre, g, x, wp, y, stats, sc, sh))
            print(f"rep {rep} pre={pre} {name:12s} {us:7.1f} us ({fl / us / 1e6:6.0f} TF/s)", flush=True)
            if hasattr(lib, "duo_diag") and rep == 2:
                d = (ctypes.c_longlong * 8)()
                try:
                    lib.duo_diag(d)
                    n = max(d[7], 1)
                    names = ["mfma", "patch wait", "transform", "barrier", "patch issue", "epilogue", "loop"]
                    print("      workgroup 0 wave 0, s_memtime ticks per tile over", n, "tiles:", ", ".join(f"{names[k]} {d[k] / n:.0f}" for k in range(7)), f"| total {sum(d[k] for k in range(7)) / n:.0f}")
                except AttributeError:
                    pass
