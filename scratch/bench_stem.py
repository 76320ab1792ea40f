import sys, os
sys.path.insert(0, os.getcwd())
import torch, ctypes
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import conv as hc, ops as hops, lib
from self_driving_model_amd.models.experts.resnet import Trunk
dev = torch.device("cuda:0")
B = 32
img = torch.randn(B, 3, 720, 1280, device=dev)
t = Trunk().to(dev).train()
for p in t.parameters(): p.requires_grad = False
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with runtime.precision(torch.float16):
    x = hops.image_to_s2d(img, torch.float16)
    cfg = hc._Cfg(t[0].spec, t[0]._packed, t[1], True, 1.0, (720, 1280))
    L = lib.get()
    g = hc.fwd_geom(t[0].spec, B, 360, 640, 16, 64, 2, orig_hw=(720, 1280))
    wp = t[0]._packed.get_fwd(t[0].weight, t[0].spec, torch.float16)
    stats = torch.zeros(16*2*64, dtype=torch.float64, device=dev)
    scale = torch.ones(64, device=dev); shift = torch.zeros(64, device=dev)
    yfull = torch.empty(B, 360, 640, 64, dtype=torch.float16, device=dev)
    ypool = torch.empty(B, 180, 320, 64, dtype=torch.float16, device=dev)
    s = hc.stream()
    print("pass1 stats-only   %.1f us" % timeit(lambda: L.am_conv_first_fused(ctypes.byref(g), 1, 1, x.data_ptr(), wp.data_ptr(), None, None, None, stats.data_ptr(), s)))
    print("pass2 bn+relu      %.1f us" % timeit(lambda: L.am_conv_first_fused(ctypes.byref(g), 1, 2, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), yfull.data_ptr(), None, s)))
    print("pass2 bn+relu+pool %.1f us" % timeit(lambda: L.am_conv_first_fused(ctypes.byref(g), 1, 3, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), ypool.data_ptr(), None, s)))
    print("maxpool            %.1f us" % timeit(lambda: hops.MaxPool3x3s2.apply(yfull)))
    print("mode0 raw+stats    %.1f us" % timeit(lambda: hc.conv_gemm(g, x, wp, None, False, yfull, stats)))
    hc.FUSE_FIRST_LAYER = True
    print("trunk fused        %.1f us" % timeit(lambda: t(x)))
    hc.FUSE_FIRST_LAYER = False
    print("trunk unfused      %.1f us" % timeit(lambda: t(x)))
