"""A/B of conv_ring16_k's 128x256 tile (AM_TUNE_RING16_M128_MIN_TILES) against what the dispatcher picked before (conv_ring_k<256,128>)
on the problems of 58-199 tiles of 256x256: correctness against torch fp32 conv on the f16-rounded operands, interleaved timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc
from self_driving_model_amd.hip import lib

L = lib.get()
KEY = lib.AM_TUNE_RING16_M128_MIN_TILES
dt = torch.float16
dev = torch.device("cuda:0")
cases = [
    ("l4 3x3 512->512 B16", 16, hc.ConvSpec(512, 512, 3, 1, 1), 23, 40),
    ("l4 3x3 512->512 B8", 8, hc.ConvSpec(512, 512, 3, 1, 1), 23, 40),
    ("l4.0 3x3 256->512 s2 B16", 16, hc.ConvSpec(256, 512, 3, 2, 1), 45, 80),
    ("l4.0 3x3 256->512 s2 B8", 8, hc.ConvSpec(256, 512, 3, 2, 1), 45, 80),
    ("l3 3x3 256->256 B16", 16, hc.ConvSpec(256, 256, 3, 1, 1), 45, 80),
    ("l3 3x3 256->256 B8", 8, hc.ConvSpec(256, 256, 3, 1, 1), 45, 80),
    ("l3.0 3x3 128->256 s2 B8", 8, hc.ConvSpec(128, 256, 3, 2, 1), 90, 160),
    ("head 3x3 512->256 B32", 32, hc.ConvSpec(512, 256, 3, 1, 1), 23, 40),
    ("head 3x3 512->256 B16", 16, hc.ConvSpec(512, 256, 3, 1, 1), 23, 40),
    ("head 3x3 512->256 B64", 64, hc.ConvSpec(512, 256, 3, 1, 1), 23, 40),
]
ROUNDS = 7
for name, B, s, IH, IW in cases:
    x = torch.randn(B, IH, IW, s.cin, device=dev).to(dt)
    w = torch.randn(s.cout, s.cin, s.k, s.k, device=dev) * 0.05
    bias = torch.randn(s.cout, device=dev)
    wp = hc.pack_fwd(w, s, dt)
    OH, OW = hc.out_size(IH, s), hc.out_size(IW, s)
    g = hc.fwd_geom(s, B, IH, IW, s.cin, s.cout, 2)
    ref = torch.nn.functional.conv2d(x[:2].float().permute(0, 3, 1, 2), w.to(dt).float(), None, s.stride, s.pad).permute(0, 2, 3, 1)
    res = {}
    for v in (1 << 30, 100):
        L.am_set_tuning(KEY, v)
        y = torch.empty(B, OH, OW, s.cout, device=dev, dtype=dt)
        stats = torch.zeros(16 * 2 * s.cout, dtype=torch.float64, device=dev)
        hc.conv_gemm(g, x, wp, None, False, y, stats)
        kid = L.am_conv_last_variant()
        yb = torch.empty_like(y)
        hc.conv_gemm(g, x, wp, bias, True, yb, None)
        torch.cuda.synchronize()
        e = float((y[:2].float() - ref).abs().max() / ref.abs().max())
        eb = float((yb[:2].float() - torch.relu(ref + bias)).abs().max() / ref.abs().max())
        res[v] = (kid, y, stats.view(16, 2, s.cout).sum(0), e, eb)
        assert e < 2e-3 and eb < 2e-3, (name, v, e, eb)
    es = float((res[100][2] - res[1 << 30][2]).abs().max() / res[1 << 30][2].abs().max())
    assert es < 1e-5, (name, es)
    times = {v: [] for v in res}
    y = torch.empty(B, OH, OW, s.cout, device=dev, dtype=dt)
    stats = torch.zeros(16 * 2 * s.cout, dtype=torch.float64, device=dev)
    n = 10
    for r in range(ROUNDS + 1):
        for v in res:
            L.am_set_tuning(KEY, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                hc.conv_gemm(g, x, wp, None, False, y, stats)
            e1.record(); torch.cuda.synchronize()
            if r:
                times[v].append(e0.elapsed_time(e1) / n * 1e3)
    fl = 2.0 * B * OH * OW * s.cin * s.k * s.k * s.cout
    M = B * OH * OW
    line = f"{name:28s} tiles256 {((M + 255) // 256) * ((s.cout + 255) // 256):4d}"
    for v in res:
        t = sorted(times[v])[len(times[v]) // 2]
        line += f" | {hc.CONV_KERNEL_NAMES.get(res[v][0], res[v][0]):28s} {t:7.1f} us {fl / t * 1e-6:7.1f} TF/s"
    print(line, flush=True)
L.am_set_tuning(KEY, 100)
