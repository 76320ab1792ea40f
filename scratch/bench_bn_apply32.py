"""bn_apply_k bandwidth on the 4a step's (B = 32) block-end shapes, tensors cycled so that nothing is served from the 256 MB MALL."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd.hip import conv as hc
L = hc._L(); dev = torch.device("cuda:0"); s = hc.stream()
p = lambda a: a.data_ptr()
for shape in ((32, 180, 320, 64), (32, 90, 160, 128), (32, 45, 80, 256), (32, 23, 40, 512)):
    C = shape[-1]; n = shape[0] * shape[1] * shape[2] * C; P = n // C; MB = n * 2 / 1e6
    K = max(2, int(1200 / (3 * MB)) + 1)  # rotate over > 1 GB of distinct tensors
    xs = [torch.randn(shape, device=dev).half() for _ in range(K)]; rs = [torch.randn(shape, device=dev).half() for _ in range(K)]; ys = [torch.empty(shape, device=dev, dtype=torch.float16) for _ in range(K)]
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev)
    def run(mode, reps=3):
        for i in range(K):
            if mode == 0: L.am_bn_apply(1, p(xs[i]), C, p(sc), p(sh), None, 0, 1, p(ys[i]), C, P, C, s)
            elif mode == 1: L.am_bn_apply(1, p(xs[i]), C, p(sc), p(sh), p(rs[i]), C, 1, p(ys[i]), C, P, C, s)
            else: L.am_bn_apply2(1, p(xs[i]), C, p(sc), p(sh), p(rs[i]), C, p(sc), p(sh), 3, p(ys[i]), C, P, C, s)
    out = []
    for mode, name, nt in ((0, "apply", 2), (1, "apply+res", 3), (2, "apply2 (residual affine)", 3)):
        run(mode); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run(mode)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * K) * 1e3
        out.append(f"{name} {us:.1f} us {nt * MB / us / 1e3:.2f} TB/s")
    print(shape, f"{MB:.0f} MB/tensor, {K} tensor sets: " + " | ".join(out), flush=True)
    del xs, rs, ys
