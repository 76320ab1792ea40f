import sys, os
sys.path.insert(0, os.getcwd())
import torch
from self_driving_model_amd.hip import conv as hc
L = hc._L(); dev = torch.device("cuda:0")
for shape in ((32, 90, 160, 128), (32, 45, 80, 256)):
    x = torch.randn(shape, device=dev).half(); r = torch.randn(shape, device=dev).half(); y = torch.empty_like(x)
    C = shape[-1]; P = x.numel() // C
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10 * 1e3
    s = hc.stream()
    print(shape, "plain res %.1f us" % t(lambda: L.am_bn_apply2(1, x.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), r.data_ptr(), C, None, None, 1, y.data_ptr(), C, P, C, s)),
          "affine res %.1f us" % t(lambda: L.am_bn_apply2(1, x.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), r.data_ptr(), C, sc.data_ptr(), sh.data_ptr(), 1, y.data_ptr(), C, P, C, s)))
