"""Fused-stem passes only (statistics pass, conv+BN+ReLU+maxpool pass), 5 launches each: target of scratch/pmc_stem.sh."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import conv as hc, ops as hops, lib
from self_driving_model_amd.models.experts.resnet import Trunk
dev = torch.device("cuda:0")
B = 32
img = torch.randn(B, 3, 720, 1280, device=dev)
t = Trunk().to(dev).train()
with runtime.precision(torch.float16):
    x = hops.image_to_s2d(img, torch.float16)
    L = lib.get()
    g = hc.fwd_geom(t[0].spec, B, 360, 640, 16, 64, 2, orig_hw=(720, 1280))
    wp = t[0]._packed.get_fwd(t[0].weight, t[0].spec, torch.float16)
    stats = torch.zeros(16 * 2 * 64, dtype=torch.float64, device=dev)
    scale = torch.ones(64, device=dev); shift = torch.zeros(64, device=dev)
    ypool = torch.empty(B, 180, 320, 64, dtype=torch.float16, device=dev)
    s = hc.stream()
    for _ in range(5):
        L.am_conv_first_fused(ctypes.byref(g), 1, 1, x.data_ptr(), wp.data_ptr(), None, None, None, stats.data_ptr(), s)
        L.am_conv_first_fused(ctypes.byref(g), 1, 3, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), ypool.data_ptr(), None, s)
    torch.cuda.synchronize()
