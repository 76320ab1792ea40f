"""Stem pooling passes at the cfg2 / cfg3 sizes: max-pool forward (plain, fused with BatchNorm + ReLU), backward (plain, with the
BatchNorm-backward sums)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc
L = hc._L(); dev = torch.device("cuda:0"); st = hc.stream(); p = lambda a: a.data_ptr()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (8, 16):
    H, W, C = 360, 640, 64
    x = torch.randn(B, H, W, C, device=dev).half(); OH, OW = 180, 320
    y = torch.empty(B, OH, OW, C, device=dev, dtype=torch.float16); arg = torch.empty(B, OH, OW, C, dtype=torch.uint8, device=dev)
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev); mean = torch.randn(C, device=dev); rstd = torch.rand(C, device=dev) + 0.5
    dy = torch.randn(B, OH, OW, C, device=dev).half(); dx = torch.empty_like(x); sums = torch.zeros(16 * 2 * C, dtype=torch.float64, device=dev)
    MB = x.numel() * 2 / 1e6
    r = {"fwd": t(lambda: L.am_maxpool3x3s2_fwd(1, p(x), p(y), p(arg), B, H, W, C, st)),
         "fwd+bn": t(lambda: L.am_bn_relu_maxpool3x3s2_fwd(1, p(x), p(sc), p(sh), p(y), p(arg), B, H, W, C, st)),
         "bwd": t(lambda: L.am_maxpool3x3s2_bwd(1, p(dy), p(arg), p(dx), B, H, W, C, st)),
         "bwd+bn sums": t(lambda: L.am_maxpool3x3s2_bwd_bn(1, p(dy), p(arg), p(dx), B, H, W, C, p(x), p(mean), p(rstd), p(sc), p(sh), p(sums), st))}
    print(f"B={B} ({MB:.0f} MB full-resolution tensor): " + "  ".join(f"{k} {v:.1f} us" for k, v in r.items()), flush=True)
