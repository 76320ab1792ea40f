"""BatchNorm pass micro-benchmark (f16): forward apply (+residual), backward reduce / apply with the ReLU mask from the activation
and from the sign of the normalised conv output; GB/s of algorithmic traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc
L = hc._L(); dev = torch.device("cuda:0"); s = hc.stream()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for shape in ((16, 180, 320, 64), (16, 90, 160, 128), (32, 45, 80, 256), (32, 23, 40, 512)):
    x = torch.randn(shape, device=dev).half(); r = torch.randn(shape, device=dev).half(); y = torch.empty_like(x); dz = torch.empty_like(x)
    dy = torch.randn(shape, device=dev).half()
    C = shape[-1]; P = x.numel() // C; MB = x.numel() * 2 / 1e6
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev); mean = torch.randn(C, device=dev); rstd = torch.rand(C, device=dev) + 0.5
    coef = torch.rand(3 * C, device=dev); sums = torch.zeros(16 * 2 * C, dtype=torch.float64, device=dev)
    p = lambda a: a.data_ptr()
    res = {
        "apply": (2, t(lambda: L.am_bn_apply(1, p(x), C, p(sc), p(sh), None, 0, 1, p(y), C, P, C, s))),
        "apply+res": (3, t(lambda: L.am_bn_apply(1, p(x), C, p(sc), p(sh), p(r), C, 1, p(y), C, P, C, s))),
        "bwd_reduce(y)": (3, t(lambda: L.am_bn_bwd_reduce(1, p(dy), C, p(y), C, p(x), C, p(mean), p(rstd), 1, p(sums), P, C, s))),
        "bwd_reduce(sign)": (2, t(lambda: L.am_bn_bwd_reduce_sign(1, p(dy), C, p(x), C, p(mean), p(rstd), p(sc), p(sh), p(sums), P, C, s))),
        "bwd_apply(y)": (4, t(lambda: L.am_bn_bwd_apply(1, p(dy), C, p(y), C, p(x), C, p(mean), p(rstd), p(coef), 1, p(dz), C, None, 0, P, C, s))),
        "bwd_apply(sign)": (3, t(lambda: L.am_bn_bwd_apply_sign(1, p(dy), C, p(x), C, p(mean), p(rstd), p(coef), p(sc), p(sh), p(dz), C, P, C, s))),
        "bwd_apply(y)+dres": (5, t(lambda: L.am_bn_bwd_apply(1, p(dy), C, p(y), C, p(x), C, p(mean), p(rstd), p(coef), 1, p(dz), C, p(r), C, P, C, s))),
    }
    print(shape, f"{MB:.0f} MB/tensor: " + "  ".join(f"{k} {us:.1f}us {n * MB / us / 1e3:.2f}TB/s" for k, (n, us) in res.items()), flush=True)
