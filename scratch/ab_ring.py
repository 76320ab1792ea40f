"""A/B of the ring kernels (AM_TUNE_RING 0 / 1 / 2) on the layer shapes of the 4a step: correctness against variant 0 and
against torch fp32 conv on the f16-rounded operands, then interleaved timing rounds in ONE process (median / min)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc
from self_driving_model_amd.hip import lib

L = lib.get()
B = int(os.environ.get("B", 32))
ROUNDS = int(os.environ.get("ROUNDS", 7))
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1,3").split(",")]
dt = torch.float16
layers = [
    ("l2 3x3 128->128", hc.ConvSpec(128, 128, 3, 1, 1), 90, 160),
    ("l3.0 3x3 128->256 s2", hc.ConvSpec(128, 256, 3, 2, 1), 90, 160),
    ("l3 3x3 256->256", hc.ConvSpec(256, 256, 3, 1, 1), 45, 80),
    ("l4.0 3x3 256->512 s2", hc.ConvSpec(256, 512, 3, 2, 1), 45, 80),
    ("l4 3x3 512->512", hc.ConvSpec(512, 512, 3, 1, 1), 23, 40),
    ("pol2 3x3 64->128 s2", hc.ConvSpec(64, 128, 3, 2, 1), 180, 320),
    ("pol3 3x3 128->256 s2", hc.ConvSpec(128, 256, 3, 2, 1), 90, 160),
    ("l3.0 ds 1x1 128->256 s2", hc.ConvSpec(128, 256, 1, 2, 0), 90, 160),
    ("l4.0 ds 1x1 256->512 s2", hc.ConvSpec(256, 512, 1, 2, 0), 45, 80),
    ("head 3x3 512->256", hc.ConvSpec(512, 256, 3, 1, 1), 23, 40),
]
if os.environ.get("ONLY"):
    layers = [l for l in layers if any(k in l[0] for k in os.environ["ONLY"].split(","))]
dev = torch.device("cuda:0")
print(f"B={B} rounds={ROUNDS}")
names = {0: "ring32", 1: "ring16", 2: "ring16s", 3: "ring16a", 4: "ring16i"}
tot = {v: 0.0 for v in VARIANTS}
for name, s, IH, IW in layers:
    x = torch.randn(B, IH, IW, s.cin, device=dev).to(dt)
    w = torch.randn(s.cout, s.cin, s.k, s.k, device=dev) * 0.05
    bias = torch.randn(s.cout, device=dev)
    wp = hc.pack_fwd(w, s, dt)
    OH, OW = hc.out_size(IH, s), hc.out_size(IW, s)
    g = hc.fwd_geom(s, B, IH, IW, s.cin, s.cout, 2)
    outs, sts, outs_b = {}, {}, {}
    for v in VARIANTS:
        L.am_set_tuning(0, v)
        y = torch.empty(B, OH, OW, s.cout, device=dev, dtype=dt)
        stats = torch.zeros(16 * 2 * s.cout, dtype=torch.float64, device=dev)
        hc.conv_gemm(g, x, wp, None, False, y, stats)
        yb = torch.empty_like(y)
        hc.conv_gemm(g, x, wp, bias, True, yb, None)
        torch.cuda.synchronize()
        outs[v], sts[v], outs_b[v] = y, stats.view(16, 2, s.cout).sum(0), yb
        kid = L.am_conv_last_variant()
        if "head" not in name and "128->128" not in name and "64->128" not in name:
            assert (kid == 1) == (v == 0) and (kid == 11) == (v > 0), (name, v, kid)
    # reference on a slice of the batch (fp32 conv on the f16-rounded operands)
    nb = 2
    ref = torch.nn.functional.conv2d(x[:nb].float().permute(0, 3, 1, 2), w.to(dt).float(), None, s.stride, s.pad).permute(0, 2, 3, 1)
    refb = torch.relu(ref + bias)
    for v in VARIANTS:
        e = float((outs[v][:nb].float() - ref).abs().max() / ref.abs().max())
        eb = float((outs_b[v][:nb].float() - refb).abs().max() / refb.abs().max())
        es = float((sts[v] - sts[VARIANTS[0]]).abs().max() / sts[VARIANTS[0]].abs().max())
        same = float((outs[v].float() - outs[VARIANTS[0]].float()).abs().max())
        assert e < 2e-3 and eb < 2e-3 and es < 1e-5, (name, v, e, eb, es)
        print(f"   check {names[v]:8s} max rel err vs fp32 conv {e:.1e} (bias+relu {eb:.1e}), stats vs v0 {es:.1e}, max |y - y_v0| {same:.2e}")
    times = {v: [] for v in VARIANTS}
    y = torch.empty(B, OH, OW, s.cout, device=dev, dtype=dt)
    stats = torch.zeros(16 * 2 * s.cout, dtype=torch.float64, device=dev)
    n = 8
    for r in range(ROUNDS + 1):
        for v in VARIANTS:
            L.am_set_tuning(0, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                hc.conv_gemm(g, x, wp, None, False, y, stats)
            e1.record(); torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) / n)
    fl = 2.0 * B * OH * OW * s.cin * s.k * s.k * s.cout
    clk = {}
    for v in VARIANTS:  # in-kernel clock and cycles per K-step of workgroup 0 (256x256 tiles only)
        L.am_set_tuning(0, v)
        hc.conv_gemm(g, x, wp, None, False, y, stats)
        c = (ctypes.c_longlong * 8)()
        L.am_diag_ring_clock(c, hc.stream())
        clk[v] = (c[0] / max(c[2], 1), c[0] / max(c[1], 1) / 10.0, c[3], (c[5], c[6], c[7], c[4]), c[0]) if L.am_conv_last_variant() in (1, 11) else None
    line = f"{name:24s} M={B*OH*OW:7d} K={s.cin*s.k*s.k:5d} N={s.cout:4d} "
    for v in VARIANTS:
        t = sorted(times[v]); med, mn = t[len(t) // 2], t[0]
        tot[v] += med
        line += f"| {names[v]} {med*1e3:7.1f} us {fl/med/1e9:6.0f} TF (min {mn*1e3:6.1f}) " + (f"[{clk[v][0]:.0f} cyc/kstep @{clk[v][1]:.2f} GHz; pro {clk[v][2]} loop {clk[v][4]} epi {clk[v][3]}] " if clk[v] else "")
    print(line, flush=True)
L.am_set_tuning(0, 0)
# head conv (115 x 2 tiles of 256x128): the ring kernel below one workgroup per CU vs the two-stage kernel
for thr in (256, 192):
    L.am_set_tuning(1, thr)
    name, s, IH, IW = [l for l in layers if "head" in l[0]][0] if any("head" in l[0] for l in layers) else (None, None, None, None)
    if name is None:
        break
    x = torch.randn(B, IH, IW, s.cin, device=dev).to(dt)
    w = torch.randn(s.cout, s.cin, s.k, s.k, device=dev) * 0.05
    wp = hc.pack_fwd(w, s, dt)
    g = hc.fwd_geom(s, B, IH, IW, s.cin, s.cout, 2)
    y = torch.empty(B, IH, IW, s.cout, device=dev, dtype=dt)
    bias = torch.randn(s.cout, device=dev)
    ts = []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            hc.conv_gemm(g, x, wp, bias, True, y, None)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 8)
    fl = 2.0 * B * IH * IW * s.cin * 9 * s.cout
    print(f"head conv, RING128_MIN_TILES={thr}: kernel {L.am_conv_last_variant()} {sorted(ts)[2]*1e3:.1f} us {fl/sorted(ts)[2]/1e9:.0f} TF")
L.am_set_tuning(1, 256)
print("sum of medians:", {names[v]: round(t, 4) for v, t in tot.items()})
