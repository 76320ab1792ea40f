"""GPU parity tests, kernel level: every HIP entry point against the torch-CPU fp32 arithmetic the
reference runs (torch.nn.functional on CPU = the oracle for single ops), through the C ABI.

Tolerance: fp32 mode rtol 1e-3 / atol 1e-5 (BASELINE.json north_star); fp16 mode (MFMA f16, fp32
accumulate) is checked at a tolerance scaled to fp16's 2^-11 rounding.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import launched_kernel

pytestmark = pytest.mark.gpu

RT, AT = 1e-3, 1e-5


def _dev():
    return torch.device("cuda:0")


def close(a, b, rtol=RT, atol=AT, what=""):
    a = a.detach().float().cpu().numpy().astype(np.float64)
    b = b.detach().float().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


def rel_err(a, b):
    a = a.detach().float().cpu().double()
    b = b.detach().float().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(x, dtype, ld=None):
    """[B,C,H,W] cpu fp32 -> device NHWC with optional channel padding."""
    B, C, H, W = x.shape
    ld = ld or C
    out = torch.zeros(B, H, W, ld, dtype=dtype)
    out[..., :C] = x.permute(0, 2, 3, 1).to(dtype)
    return out.to(_dev())


def nchw(y, C):
    return y[..., :C].permute(0, 3, 1, 2).float().cpu()


CONV_CASES = [
    # cin, cout, k, stride, pad, H, W
    (64, 64, 3, 1, 1, 13, 20),
    (64, 128, 3, 2, 1, 23, 40),
    (64, 128, 1, 2, 0, 23, 40),
    (128, 256, 3, 2, 1, 12, 20),
    (256, 512, 3, 1, 1, 5, 7),
    (3, 64, 7, 2, 3, 45, 64),
    (3, 32, 5, 2, 2, 31, 50),
    (256, 14, 1, 1, 0, 6, 9),
    (32, 64, 3, 2, 1, 22, 33),
]


# (case, dtype) -> kernel the forward must launch at these small sizes: the exact-fp32 register-staged kernel in fp32 mode; in f16
# the two-stage LDS-DMA kernel, except the 3-channel first layers (below the size the space-to-depth patch kernel takes)
FWD_KERNEL = {(c, torch.float32): "conv_gemm_k" for c in CONV_CASES}
FWD_KERNEL.update({(c, torch.float16): ("conv_gemm_k" if c[0] == 3 else "conv_gemm2_k") for c in CONV_CASES})


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd_vs_torch(case, dtype):
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import ops as hops
    cin, cout, k, st, pad, H, W = case
    B = 3
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(cin != 3), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.relu(F.conv2d(xr, wr, br, stride=st, padding=pad))
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe).sum().backward()

    spec = hc.ConvSpec(cin, cout, k, st, pad, first=(cin == 3))
    with runtime.precision(dtype, 1.0):
        if cin == 3:
            xd = hops.image_to_nhwc(x.to(_dev()), dtype)
        else:
            xd = nhwc(x, dtype).requires_grad_()
        wd, bd = w.to(_dev()).requires_grad_(), b.to(_dev()).requires_grad_()
        cfg = hc._Cfg(spec, hc.PackedWeights(), None, True, 1.0)
        y = hc.conv_bn_act(xd, wd, bd, None, True, None, cfg, False)
        launched_kernel(FWD_KERNEL.get((case, dtype)), what=f"conv_fwd {case} {dtype}")
        es = 2 if dtype == torch.float16 else 4
        ld = hc.channel_ld(cout, es)
        assert y.shape == (B, yr.shape[2], yr.shape[3], ld)
        (y[..., :cout].float() * probe.permute(0, 2, 3, 1).to(_dev())).sum().backward()
    tol = dict(rtol=RT, atol=AT) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    close(nchw(y, cout), yr, what="conv fwd", **tol)
    if dtype == torch.float32:
        close(wd.grad, wr.grad, rtol=RT, atol=1e-4, what="wgrad")
        close(bd.grad, br.grad, rtol=RT, atol=1e-4, what="bias grad")
        if cin != 3:
            close(nchw(xd.grad, cin), xr.grad, what="dgrad")
    else:
        # fp16: inputs/weights rounded to 2^-11 and ReLU masks that flip on near-zero outputs
        assert rel_err(wd.grad, wr.grad) < 4e-2
        assert rel_err(bd.grad, br.grad) < 4e-2
        if cin != 3:
            assert rel_err(nchw(xd.grad, cin), xr.grad) < 4e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("with_res,conv_bias,sign_mask", [(False, False, True), (False, False, False), (True, False, True), (False, True, True)])
def test_conv_bn_relu_train_vs_torch(dtype, with_res, conv_bias, sign_mask):
    """conv -> BatchNorm2d(train) -> (+residual) -> ReLU, forward, running stats, all gradients.  sign_mask: backward takes the
    ReLU mask from the sign of the normalised conv output (the *_sign entries; layers without a residual) or from the activation."""
    import torch.nn as nn
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    hc.SIGN_RELU_MASK = sign_mask
    B, cin, cout, H, W = 4, 64, 128, 9, 14
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / 24.0
    cb = torch.randn(cout, generator=g) * 0.5 if conv_bias else None
    res = torch.randn(B, cout, H, W, generator=g) if with_res else None
    bn_ref = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn_ref.weight.copy_(1 + 0.2 * torch.randn(cout, generator=g))
        bn_ref.bias.copy_(0.2 * torch.randn(cout, generator=g))
        bn_ref.running_mean.copy_(0.1 * torch.randn(cout, generator=g))
        bn_ref.running_var.copy_(0.5 + torch.rand(cout, generator=g))
    bn_hip = nn.BatchNorm2d(cout)
    bn_hip.load_state_dict(bn_ref.state_dict())
    bn_hip.to(_dev())
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    cbr = cb.clone().requires_grad_() if conv_bias else None
    rr = res.clone().requires_grad_() if with_res else None
    z = bn_ref(F.conv2d(xr, wr, cbr, padding=1))
    yr = F.relu(z + rr if with_res else z)
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe).sum().backward()
    with runtime.precision(dtype, 1.0):
        xd = nhwc(x, dtype).requires_grad_()
        wd = w.to(_dev()).requires_grad_()
        cbd = cb.to(_dev()).requires_grad_() if conv_bias else None
        rd = nhwc(res, dtype).requires_grad_() if with_res else None
        cfg = hc._Cfg(hc.ConvSpec(cin, cout, 3, 1, 1), hc.PackedWeights(), bn_hip, True, 1.0)
        y = hc.ConvBnAct.apply(xd, wd, cbd, bn_hip.weight, bn_hip.bias, rd, cfg, True)
        (y.float() * probe.permute(0, 2, 3, 1).to(_dev())).sum().backward()
    if dtype == torch.float32:
        close(nchw(y, cout), yr, what="y")
        close(bn_hip.running_mean, bn_ref.running_mean, what="running_mean")
        close(bn_hip.running_var, bn_ref.running_var, what="running_var")
        hc.flush_bn_counters()
        assert int(bn_hip.num_batches_tracked) == 1
        close(nchw(xd.grad, cin), xr.grad, rtol=RT, atol=1e-4, what="dx")
        close(wd.grad, wr.grad, rtol=RT, atol=2e-4, what="dw")
        close(bn_hip.weight.grad, bn_ref.weight.grad, rtol=RT, atol=2e-4, what="dgamma")
        close(bn_hip.bias.grad, bn_ref.bias.grad, rtol=RT, atol=2e-4, what="dbeta")
        if with_res:
            close(nchw(rd.grad, cout), rr.grad, what="dres")
        if conv_bias:
            close(cbd.grad, cbr.grad, rtol=RT, atol=2e-4, what="conv bias grad (exactly 0 under train-mode BN)")
    else:
        assert rel_err(nchw(y, cout), yr) < 3e-3
        assert rel_err(bn_hip.running_var, bn_ref.running_var) < 1e-3
        assert rel_err(nchw(xd.grad, cin), xr.grad) < 4e-2
        assert rel_err(wd.grad, wr.grad) < 4e-2
        assert rel_err(bn_hip.weight.grad, bn_ref.weight.grad) < 4e-2
    hc.SIGN_RELU_MASK = True


def test_bn_eval_mode_vs_torch():
    import torch.nn as nn
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    B, cin, cout, H, W = 2, 64, 64, 7, 9
    g = torch.Generator().manual_seed(6)
    x, w = torch.randn(B, cin, H, W, generator=g), torch.randn(cout, cin, 3, 3, generator=g) / 24.0
    bn_ref = nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        bn_ref.running_mean.copy_(0.1 * torch.randn(cout, generator=g))
        bn_ref.running_var.copy_(0.5 + torch.rand(cout, generator=g))
    bn_hip = nn.BatchNorm2d(cout).eval()
    bn_hip.load_state_dict(bn_ref.state_dict())
    bn_hip.to(_dev())
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.relu(bn_ref(F.conv2d(xr, wr, padding=1)))
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe).sum().backward()
    with runtime.precision(torch.float32, 1.0):
        xd, wd = nhwc(x, torch.float32).requires_grad_(), w.to(_dev()).requires_grad_()
        cfg = hc._Cfg(hc.ConvSpec(cin, cout, 3, 1, 1), hc.PackedWeights(), bn_hip, True, 1.0)
        y = hc.ConvBnAct.apply(xd, wd, None, bn_hip.weight, bn_hip.bias, None, cfg, False)
        (y * probe.permute(0, 2, 3, 1).to(_dev())).sum().backward()
    close(nchw(y, cout), yr)
    close(nchw(xd.grad, cin), xr.grad, atol=1e-4)
    close(wd.grad, wr.grad, atol=2e-4)
    close(bn_hip.weight.grad, bn_ref.weight.grad, atol=2e-4)
    close(bn_hip.bias.grad, bn_ref.bias.grad, atol=2e-4)
    assert int(bn_hip.num_batches_tracked) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_maxpool_gap_vs_torch(dtype):
    from self_driving_model_amd.hip import ops as hops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 23, 37, generator=g)
    x[0, :, 4:6, 4:6] = 1.5  # ties inside windows: first max in scan order must win
    xq = x.to(dtype).float()
    xr = xq.clone().requires_grad_()
    yr = F.max_pool2d(xr, 3, 2, 1)
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe).sum().backward()
    xd = nhwc(xq, dtype).requires_grad_()
    y = hops.MaxPool3x3s2.apply(xd)
    (y.float() * probe.permute(0, 2, 3, 1).to(_dev())).sum().backward()
    close(nchw(y, 64), yr, rtol=0, atol=0, what="maxpool fwd is exact")
    close(nchw(xd.grad, 64), xr.grad, rtol=5e-3 if dtype == torch.float16 else 1e-6, atol=4e-3 if dtype == torch.float16 else 1e-6)
    # even sizes / a single row of blocks (the backward owns 2x2 input blocks: sizes where the last block is whole, and H = 2)
    for shp in ((1, 64, 24, 40), (1, 64, 2, 6)):
        xe = torch.randn(*shp, generator=g).to(dtype).float()
        xe[0, :, 0:2, 2:4] = 0.25
        xer = xe.clone().requires_grad_()
        ye = F.max_pool2d(xer, 3, 2, 1)
        pe = torch.randn(ye.shape, generator=g)
        (ye * pe).sum().backward()
        xed = nhwc(xe, dtype).requires_grad_()
        yd = hops.MaxPool3x3s2.apply(xed)
        (yd.float() * pe.permute(0, 2, 3, 1).to(_dev())).sum().backward()
        close(nchw(yd, 64), ye, rtol=0, atol=0, what=f"maxpool fwd {shp}")
        close(nchw(xed.grad, 64), xer.grad, rtol=5e-3 if dtype == torch.float16 else 1e-6, atol=4e-3 if dtype == torch.float16 else 1e-6, what=f"maxpool bwd {shp}")
    # GAP NHWC
    xr2 = xq.clone().requires_grad_()
    pr = xr2.mean(dim=(2, 3))
    p2 = torch.randn(pr.shape, generator=g)
    (pr * p2).sum().backward()
    xd2 = nhwc(xq, dtype).requires_grad_()
    p = hops.GapNhwc.apply(xd2, 1.0)
    (p * p2.to(_dev())).sum().backward()
    close(p, pr, rtol=1e-4, atol=1e-5)
    close(nchw(xd2.grad, 64), xr2.grad, rtol=2e-3, atol=1e-6)
    # GAP over NCHW planes
    xr3 = x.clone().requires_grad_()
    q = F.adaptive_avg_pool2d(xr3, 1).flatten(1)
    (q * p2).sum().backward()
    xd3 = x.to(_dev()).requires_grad_()
    qd = hops.GapPlane.apply(xd3)
    (qd * p2.to(_dev())).sum().backward()
    close(qd, q, rtol=1e-4, atol=1e-6)
    close(xd3.grad, xr3.grad, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("shape", [(2, 3, 23, 40, 720, 1280), (2, 19, 5, 7, 37, 50), (1, 14, 2, 3, 64, 96)])
def test_bilinear_vs_torch(shape):
    from self_driving_model_amd.hip import ops as hops
    B, C, h, w, H, W = shape
    g = torch.Generator().manual_seed(8)
    low = torch.randn(B, C, h, w, generator=g)
    lr = low.clone().requires_grad_()
    yr = F.interpolate(lr, size=(H, W), mode="bilinear", align_corners=False)
    probe = torch.randn(yr.shape, generator=g)
    (yr * probe).sum().backward()
    ld = 32 if C % 4 else C
    ld = max(ld, 16)
    ld = 32
    ldv = nhwc(low, torch.float32, ld).requires_grad_()
    y = hops.BilinearUp.apply(ldv, C, H, W, 1.0)
    (y * probe.to(_dev())).sum().backward()
    close(y, yr, rtol=1e-4, atol=1e-5)
    close(nchw(ldv.grad, C), lr.grad, rtol=1e-3, atol=1e-3)
    assert float(ldv.grad[..., C:].abs().max()) == 0.0


def test_ce2d_vs_torch():
    from self_driving_model_amd.hip import ops as hops
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 2, 19, 33, 47
    logits = torch.randn(B, C, H, W, generator=g) * 3
    tgt = torch.randint(0, C, (B, H, W), generator=g)
    tgt[torch.rand(B, H, W, generator=g) < 0.1] = 255
    lr = logits.clone().requires_grad_()
    loss_r = F.cross_entropy(lr, tgt, ignore_index=255)
    (loss_r * 1.7).backward()
    ld = logits.to(_dev()).requires_grad_()
    loss = hops.CrossEntropy2d.apply(ld, tgt.to(_dev()), 255)
    (loss * 1.7).backward()
    close(loss, loss_r, rtol=1e-5, atol=1e-6)
    close(ld.grad, lr.grad, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("C", [3, 19])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_upsample_cross_entropy_fused_vs_torch(C, dtype):
    """CrossEntropyLoss(ignore_index=255)(F.interpolate(low, bilinear), target) and its gradient with respect to the low-resolution
    logits from ONE pass over the labels (am_upsample_ce2d_*): against torch-CPU fp32 on the same (dtype-rounded) logits, against
    the two-op sequence (BilinearUp -> CrossEntropy2d), run twice for bit-identical results (no atomics on the gradient), on a
    non-integer scale, the experts' 23x40 -> 720x1280 geometry, a padded pixel stride, and an all-ignored target (nan as torch)."""
    from self_driving_model_amd.hip import ops as hops
    g = torch.Generator().manual_seed(90 + C)
    for (B, h, w, H, W, ld) in [(2, 5, 7, 37, 50, C), (3, 4, 9, 128, 288, 32), (2, 23, 40, 720, 1280, 32 if C > 3 else 8)]:
        low = (torch.randn(B, C, h, w, generator=g) * 2).to(dtype).float()
        tgt = torch.randint(0, C, (B, H, W), generator=g)
        tgt[torch.rand(B, H, W, generator=g) < 0.15] = 255
        lr = low.clone().requires_grad_()
        loss_r = F.cross_entropy(F.interpolate(lr, size=(H, W), mode="bilinear", align_corners=False), tgt, ignore_index=255)
        (loss_r * 1.3).backward()
        ref_grad = lr.grad.permute(0, 2, 3, 1)
        ls = 64.0 if dtype == torch.float16 else 1.0

        def run(fused):
            lowd = torch.zeros(B, h, w, ld, dtype=dtype, device=_dev())
            lowd[..., :C] = low.permute(0, 2, 3, 1).to(dtype).to(_dev())
            lowd.requires_grad_()
            if fused:
                loss = hops.UpsampleCrossEntropy.apply(lowd, tgt.to(_dev()), C, H, W, 255, ls)
            else:
                loss = hops.CrossEntropy2d.apply(hops.BilinearUp.apply(lowd, C, H, W, ls), tgt.to(_dev()), 255)
            (loss * 1.3).backward()
            torch.cuda.synchronize()
            return loss.detach(), lowd.grad.float() / ls

        loss, grad = run(True)
        close(loss, loss_r, rtol=1e-5, atol=1e-6)
        if dtype == torch.float32:
            close(grad[..., :C], ref_grad, rtol=1e-3, atol=1e-7)
        else:  # the f16 gradient tensor carries 11 bits
            assert rel_err(grad[..., :C], ref_grad) < 1e-3
        assert float(grad[..., C:].abs().sum()) == 0.0
        loss2, grad2 = run(True)
        assert torch.equal(grad, grad2), "fused gradient is not deterministic"
        loss_u, grad_u = run(False)
        close(loss, loss_u, rtol=1e-5, atol=1e-6)
        assert rel_err(grad, grad_u) < (1e-5 if dtype == torch.float32 else 1e-3)
    tgt_all = torch.full((1, 16, 24), 255, dtype=torch.int64)
    lowd = torch.randn(1, 2, 3, C, device=_dev()).to(dtype).requires_grad_()
    assert torch.isnan(hops.UpsampleCrossEntropy.apply(lowd, tgt_all.to(_dev()), C, 16, 24, 255, 1.0))


def test_mlp_tail_ops_vs_torch():
    import torch.nn as nn
    from self_driving_model_amd.hip import ops as hops
    g = torch.Generator().manual_seed(10)
    for (M, K, N, relu) in [(4, 4, 32, True), (32, 768, 512, True), (5, 19, 512, False), (64, 896, 128, True), (3, 128, 3, False)]:
        x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / np.sqrt(K), torch.randn(N, generator=g)
        xr, Wr, br = x.clone().requires_grad_(), W.clone().requires_grad_(), b.clone().requires_grad_()
        yr = F.linear(xr, Wr, br)
        yr = F.relu(yr) if relu else yr
        probe = torch.randn(M, N, generator=g)
        (yr * probe).sum().backward()
        xd, Wd, bd = (t.to(_dev()).requires_grad_() for t in (x, W, b))
        y = hops.LinearAct.apply(xd, Wd, bd, relu)
        (y * probe.to(_dev())).sum().backward()
        close(y, yr, rtol=1e-4, atol=1e-5)
        close(xd.grad, xr.grad, rtol=1e-4, atol=1e-5)
        close(Wd.grad, Wr.grad, rtol=1e-4, atol=1e-5)
        close(bd.grad, br.grad, rtol=1e-4, atol=1e-5)
    # LayerNorm
    for (M, D) in [(4, 64), (33, 256)]:
        x = torch.randn(M, D, generator=g) * 2 + 0.5
        ln = nn.LayerNorm(D)
        with torch.no_grad():
            ln.weight.copy_(1 + 0.1 * torch.randn(D, generator=g))
            ln.bias.copy_(0.1 * torch.randn(D, generator=g))
        xr = x.clone().requires_grad_()
        yr = ln(xr)
        probe = torch.randn(M, D, generator=g)
        (yr * probe).sum().backward()
        xd = x.to(_dev()).requires_grad_()
        gd, bd = ln.weight.detach().to(_dev()).requires_grad_(), ln.bias.detach().to(_dev()).requires_grad_()
        y = hops.LayerNormFn.apply(xd, gd, bd, ln.eps)
        (y * probe.to(_dev())).sum().backward()
        close(y, yr, rtol=1e-4, atol=1e-5)
        close(xd.grad, xr.grad, rtol=1e-4, atol=1e-5)
        close(gd.grad, ln.weight.grad, rtol=1e-4, atol=1e-5)
        close(bd.grad, ln.bias.grad, rtol=1e-4, atol=1e-5)
    # dropout: mask statistics and gradient consistency
    x = torch.ones(64, 512, device=_dev(), requires_grad=True)
    y = hops.DropoutFn.apply(x, 0.1)
    keep = float((y > 0).float().mean())
    assert abs(keep - 0.9) < 0.01
    assert torch.allclose(y[y > 0], torch.full_like(y[y > 0], 1 / 0.9))
    y.sum().backward()
    assert torch.equal(x.grad > 0, y.detach() > 0)


@pytest.mark.parametrize("use_softmax,temp,topk,E", [(True, 1.0, 0, 3), (True, 0.5, 0, 4), (False, 1.0, 0, 3), (True, 1.0, 2, 4), (False, 1.0, 2, 4)])
def test_gate_combine_vs_torch(use_softmax, temp, topk, E):
    from self_driving_model_amd.hip import ops as hops
    g = torch.Generator().manual_seed(11)
    B, D = 6, 256
    logits = torch.randn(B, E, generator=g)
    procs = [torch.randn(B, D, generator=g) for _ in range(E)]
    lr = logits.clone().requires_grad_()
    pr = [p.clone().requires_grad_() for p in procs]
    lg = lr
    if topk:
        vals, idx = torch.topk(lg, topk, dim=1)
        lg = torch.full_like(lg, float("-inf")).scatter(1, idx, vals)
    if use_softmax:
        wr = F.softmax(lg / temp, dim=1)
    else:
        s = torch.sigmoid(lg)
        wr = s / (s.sum(dim=1, keepdim=True) + 1e-8)
    cr = sum(wr[:, i:i + 1] * pr[i] for i in range(E))
    p1, p2 = torch.randn(B, D, generator=g), torch.randn(B, E, generator=g)
    ((cr * p1).sum() + (wr * p2).sum()).backward()
    ld = logits.to(_dev()).requires_grad_()
    pd = [p.to(_dev()).requires_grad_() for p in procs]
    w, c = hops.GateCombine.apply(ld, temp, use_softmax, topk, *pd)
    ((c * p1.to(_dev())).sum() + (w * p2.to(_dev())).sum()).backward()
    close(w, wr, rtol=1e-5, atol=1e-6)
    close(c, cr, rtol=1e-5, atol=1e-5)
    close(ld.grad, lr.grad, rtol=1e-4, atol=1e-5)
    for a, b in zip(pd, pr):
        close(a.grad, b.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(w.sum(dim=1).cpu(), torch.ones(B), atol=1e-6) and bool((w >= 0).all())


def test_adamw_and_clip_vs_torch():
    from self_driving_model_amd.hip import lib
    from self_driving_model_amd.hip.conv import ptr, stream
    L = lib.get()
    g = torch.Generator().manual_seed(12)
    n = 100_003
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) * 0.01 for _ in range(3)]
    pr = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([pr], lr=4e-4, weight_decay=1e-4)
    pd = p0.to(_dev())
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step, gr in enumerate(grads, 1):
        pr.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([pr], max_norm=1.0)
        opt.step()
        gd = gr.to(_dev())
        acc = torch.zeros(1, dtype=torch.float64, device=_dev())
        L.am_sumsq_accumulate(ptr(gd), n, ptr(acc), stream())
        L.am_adamw_step(ptr(pd), ptr(gd), ptr(m), ptr(v), n, 4e-4, 0.9, 0.999, 1e-8, 1e-4, step, 1.0, ptr(acc), None, stream())
        close(acc.sqrt(), gr.double().norm(), rtol=1e-6, atol=0)
    close(pd, pr, rtol=1e-5, atol=1e-6)
    # non-finite norm skips the step and counts it
    skipped = torch.zeros(1, dtype=torch.int32, device=_dev())
    acc = torch.full((1,), float("inf"), dtype=torch.float64, device=_dev())
    before = pd.clone()
    L.am_adamw_step(ptr(pd), ptr(gd), ptr(m), ptr(v), n, 4e-4, 0.9, 0.999, 1e-8, 1e-4, 4, 1.0, ptr(acc), ptr(skipped), stream())
    assert torch.equal(pd, before) and int(skipped) == 1


WS_KERNEL = {(2, 180, 320): "conv3x3_c64n64_duo_k", (1, 200, 333): "conv3x3_c64n64_duo_k",
             (3, 45, 64): "conv_gemm2_k"}  # the small problem stays below the weights-in-registers kernel's size gate


@pytest.mark.parametrize("mfma16", [1, 0])
@pytest.mark.parametrize("shape", [(2, 180, 320), (3, 45, 64), (1, 200, 333)])
def test_conv3x3_weights_stationary_kernel_vs_torch(shape, mfma16, request):
    """The 64->64 3x3 weights-in-registers kernel (conv_patch3.hip) takes over from the gather-GEMM for large fp16 problems: same
    results, including BatchNorm statistics that must exclude the out-of-image rows of edge tiles; in both of its MFMA forms
    (AM_TUNE_DUO_MFMA16: 16x16x32 over 16-pixel row fragments at a 160-byte patch pitch -- the default -- and 32x32x16 at 144)."""
    from self_driving_model_amd.hip import conv as hc
    old_form = hc._L().am_set_tuning(8, mfma16)
    request.addfinalizer(lambda: hc._L().am_set_tuning(8, old_form))
    B, H, W = shape
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, 64, H, W, generator=g).half().float()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().float()
    yr = F.conv2d(x, w, padding=1)
    s = hc.ConvSpec(64, 64, 3, 1, 1)
    xd = nhwc(x, torch.float16)
    wp = hc.pack_fwd(w.to(_dev()), s, torch.float16)
    y = torch.zeros(B, H, W, 64, dtype=torch.float16, device=_dev())
    stats = torch.zeros(16 * 2 * 64, dtype=torch.float64, device=_dev())
    geom = hc.fwd_geom(s, B, H, W, 64, 64, 2)
    hc.conv_gemm(geom, xd, wp, None, False, y, stats)
    launched_kernel(WS_KERNEL.get(shape), what=f"c64n64 {shape}")
    torch.cuda.synchronize()
    close(nchw(y, 64), yr, rtol=2e-3, atol=2e-3)
    st = stats.view(16, 2, 64).sum(0).cpu()
    np.testing.assert_allclose(st[0].numpy(), yr.double().sum(dim=(0, 2, 3)).numpy(), rtol=1e-3, atol=0.5)
    np.testing.assert_allclose(st[1].numpy(), (yr.double() ** 2).sum(dim=(0, 2, 3)).numpy(), rtol=1e-3)


@pytest.mark.parametrize("case", [(128, 128, 2, 90, 160), (64, 128, 3, 37, 50), (128, 96, 2, 41, 77), (256, 128, 1, 64, 96)])
def test_conv_halo_kernel_vs_torch_and_ring_kernel(case):
    """3x3 / stride-1 layers with 64 < N <= 128 (ResNet layer2 and its input gradient): the halo-staged kernel conv_halo_k
    (conv_halo.hip; forced here for small grids through AM_TUNE_HALO_MIN_TILES) against torch and, bit for bit in the output
    and to fp64 rounding in the statistics, against the gather kernels it replaces.  Edge tiles in both directions (the
    BatchNorm statistics must exclude tile pixels outside the image), Cin of 2 / 4 / 8 chunks, N below the tile width, the
    input-gradient geometry (dgrad_plans + pack_dgrad), bias + ReLU and residual epilogues."""
    import ctypes
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import lib
    L = lib.get()
    cin, cout, B, H, W = case
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).half().float()
    b = torch.randn(cout, generator=g)
    r = torch.randn(B, cout, H, W, generator=g).half().float()
    s = hc.ConvSpec(cin, cout, 3, 1, 1)
    ldo = hc.channel_ld(cout, 2)
    geom = hc.fwd_geom(s, B, H, W, cin, ldo, 2)
    xd, wp = nhwc(x, torch.float16), hc.pack_fwd(w.to(_dev()), s, torch.float16)
    yr = F.conv2d(x, w, padding=1)
    outs = {}
    for min_tiles, kernel in ((1, "conv_halo_k"), (1 << 30, None)):
        old = L.am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, min_tiles)
        try:
            y = torch.zeros(B, H, W, ldo, dtype=torch.float16, device=_dev())
            stats = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=_dev())
            hc.conv_gemm(geom, xd, wp, None, False, y, stats)
            name = launched_kernel(kernel, what=f"halo {case}")
            assert kernel is not None or name != "conv_halo_k"
            y2 = torch.zeros_like(y)
            hc.conv_gemm(geom, xd, wp, b.to(_dev()), True, y2, None)
            y3 = torch.zeros_like(y)
            if ldo == cout:
                try:
                    L.am_conv_gemm_res(ctypes.byref(geom), hc.dt_code(torch.float16), hc.ptr(xd), hc.ptr(wp), hc.ptr(b.to(_dev())),
                                       hc.ptr(nhwc(r, torch.float16)), 1, hc.ptr(y3), hc.stream())
                    # bit for bit the two-pass sequence on the same kernel: conv + bias rounded to f16, then + residual, ReLU, rounded
                    y1 = torch.zeros_like(y)
                    hc.conv_gemm(geom, xd, wp, b.to(_dev()), False, y1, None)
                    torch.cuda.synchronize()
                    assert torch.equal(y3, torch.relu(y1.float() + nhwc(r, torch.float16).float()).half()), "residual epilogue differs from the two-pass sequence"
                except RuntimeError as e:  # the gather kernels decline small grids: only the halo run is checked then
                    assert "UNSUPPORTED" in str(e) and kernel is None
                    y3 = None
            # input gradient of the same layer: dX = conv(dY, flipped weights), the stride-1 plan has the forward geometry
            (gd, taps), = hc.dgrad_plans(s, B, H, W, cin, ldo, 2)
            dyd = nhwc(r, torch.float16, ld=ldo)
            dx = torch.zeros(B, H, W, cin, dtype=torch.float16, device=_dev())
            hc.conv_gemm(gd, dyd, hc.pack_dgrad(w.to(_dev()), taps, torch.float16, ldo), None, False, dx, None)
            dg_name = launched_kernel(None, what=f"halo dgrad {case}")
            torch.cuda.synchronize()
            outs[min_tiles] = (y, stats.view(16, 2, cout).sum(0).cpu(), y2, y3, dx, dg_name)
        finally:
            L.am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, old)
    y, st, y2, y3, dx, dg_name = outs[1]
    assert (dg_name == "conv_halo_k") == (64 < cin <= 128), dg_name  # the dgrad's N is the layer's Cin
    close(nchw(y, cout), yr, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(st[0].numpy(), yr.double().sum(dim=(0, 2, 3)).numpy(), rtol=1e-3, atol=0.5)
    np.testing.assert_allclose(st[1].numpy(), (yr.double() ** 2).sum(dim=(0, 2, 3)).numpy(), rtol=1e-3)
    close(nchw(y2, cout), F.relu(yr + b.view(1, -1, 1, 1)), rtol=3e-3, atol=3e-3)
    if y3 is not None:
        close(nchw(y3, cout), F.relu(yr + b.view(1, -1, 1, 1) + r), rtol=3e-3, atol=4e-3)
    dxr = F.conv_transpose2d(r, w, padding=1)
    close(nchw(dx, cin), dxr, rtol=3e-3, atol=3e-3)
    # same products, same fp32 accumulation order over K (tap-major inside a chunk vs chunk-major inside a tap differ): close, not equal
    yo, sto, y2o, y3o, dxo, _ = outs[1 << 30]
    assert rel_err(y, yo) < 2e-3 and rel_err(dx, dxo) < 2e-3
    np.testing.assert_allclose(st.numpy(), sto.numpy(), rtol=1e-4, atol=0.05)
    assert float(y[..., cout:].abs().sum()) == 0.0  # pad channels of the pixel stride stay zero


@pytest.mark.parametrize("case", [(256, 256, 2, 45, 80), (512, 512, 2, 23, 40), (512, 256, 3, 23, 40), (256, 256, 2, 18, 37), (128, 256, 1, 31, 120),
                                  (64, 512, 2, 15, 16)])
def test_conv_band16_kernel_vs_torch_and_ring_kernel(case):
    """3x3 / stride-1 layers with N = 256 / 512 on maps up to 128 pixels wide (ResNet layers 3-4, the 512 -> 256 heads, their
    input gradients): the row-band halo kernel conv_band16_k (conv_band16.hip; forced for small grids through
    AM_TUNE_BAND_MIN_TILES) against torch and against the gather kernels it replaces.  Covered: the two 720p shapes (45 x 80:
    three-row bands; 23 x 40: six-row bands whose 16-pixel fragments wrap from one band row to the next, last band with a row
    below the image), widths that are no multiple of 8, a two-row band at the maximum width, a band taller than the image,
    Cin of 2 / 4 / 8 / 16 chunks, one and two N tiles, BatchNorm statistics (dead tile rows excluded), bias + ReLU and residual
    epilogues, the input-gradient geometry."""
    import ctypes
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import lib
    L = lib.get()
    cin, cout, B, H, W = case
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).half().float()
    b = torch.randn(cout, generator=g)
    r = torch.randn(B, cout, H, W, generator=g).half().float()
    s = hc.ConvSpec(cin, cout, 3, 1, 1)
    geom = hc.fwd_geom(s, B, H, W, cin, cout, 2)
    xd, wp = nhwc(x, torch.float16), hc.pack_fwd(w.to(_dev()), s, torch.float16)
    yr = F.conv2d(x, w, padding=1)
    outs = {}
    for min_tiles, kernel in ((1, "conv_band16_k"), (1 << 30, None)):
        old = L.am_set_tuning(lib.AM_TUNE_BAND_MIN_TILES, min_tiles)
        try:
            y = torch.zeros(B, H, W, cout, dtype=torch.float16, device=_dev())
            stats = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=_dev())
            hc.conv_gemm(geom, xd, wp, None, False, y, stats)
            name = launched_kernel(kernel, what=f"band16 {case}")
            assert kernel is not None or name != "conv_band16_k"
            y2 = torch.zeros_like(y)
            hc.conv_gemm(geom, xd, wp, b.to(_dev()), True, y2, None)
            y3 = torch.zeros_like(y)
            try:
                L.am_conv_gemm_res(ctypes.byref(geom), hc.dt_code(torch.float16), hc.ptr(xd), hc.ptr(wp), hc.ptr(b.to(_dev())),
                                   hc.ptr(nhwc(r, torch.float16)), 1, hc.ptr(y3), hc.stream())
                y1 = torch.zeros_like(y)
                hc.conv_gemm(geom, xd, wp, b.to(_dev()), False, y1, None)
                torch.cuda.synchronize()
                assert torch.equal(y3, torch.relu(y1.float() + nhwc(r, torch.float16).float()).half()), "residual epilogue differs from the two-pass sequence"
            except RuntimeError as e:
                assert "UNSUPPORTED" in str(e) and kernel is None
                y3 = None
            (gd, taps), = hc.dgrad_plans(s, B, H, W, cin, cout, 2)
            dx = torch.zeros(B, H, W, cin, dtype=torch.float16, device=_dev())
            hc.conv_gemm(gd, nhwc(r, torch.float16), hc.pack_dgrad(w.to(_dev()), taps, torch.float16, cout), None, False, dx, None)
            dg_name = launched_kernel(None, what=f"band16 dgrad {case}")
            torch.cuda.synchronize()
            outs[min_tiles] = (y, stats.view(16, 2, cout).sum(0).cpu(), y2, y3, dx, dg_name)
        finally:
            L.am_set_tuning(lib.AM_TUNE_BAND_MIN_TILES, old)
    y, st, y2, y3, dx, dg_name = outs[1]
    assert (dg_name == "conv_band16_k") == (cin % 256 == 0), dg_name  # the dgrad's N is the layer's Cin
    close(nchw(y, cout), yr, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(st[0].numpy(), yr.double().sum(dim=(0, 2, 3)).numpy(), rtol=1e-3, atol=0.5)
    np.testing.assert_allclose(st[1].numpy(), (yr.double() ** 2).sum(dim=(0, 2, 3)).numpy(), rtol=1e-3)
    close(nchw(y2, cout), F.relu(yr + b.view(1, -1, 1, 1)), rtol=3e-3, atol=3e-3)
    if y3 is not None:
        close(nchw(y3, cout), F.relu(yr + b.view(1, -1, 1, 1) + r), rtol=3e-3, atol=4e-3)
    close(nchw(dx, cin), F.conv_transpose2d(r, w, padding=1), rtol=3e-3, atol=4e-3)
    yo, sto, y2o, y3o, dxo, _ = outs[1 << 30]
    assert rel_err(y, yo) < 2e-3 and rel_err(dx, dxo) < 2e-3 and rel_err(y2, y2o) < 2e-3
    np.testing.assert_allclose(st.numpy(), sto.numpy(), rtol=1e-4, atol=0.05)


@pytest.mark.parametrize("spec", [(64, 7, 3), (32, 5, 2)])
def test_first_layer_s2d_patch_kernel_and_fused_bn_relu(spec):
    """Large first-layer problems run the weights-stationary s2d kernel (conv_s2d.hip).  Mode 0 (raw + statistics) is
    checked against torch; the frozen-stem two-pass path (statistics-only pass, then conv+BN+ReLU in the epilogue) must
    equal the unfused sequence conv -> BatchNorm2d(train) -> ReLU, including the running-statistics update."""
    import torch.nn as nn
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import ops as hops
    cout, k, pad = spec
    B, H, W = 2, 390, 518  # odd halves exercise ragged edge tiles: out 195 x 259
    g = torch.Generator().manual_seed(k)
    img = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(cout, 3, k, k, generator=g) / (3 * k * k) ** 0.5
    bn_ref = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn_ref.weight.copy_(1 + 0.2 * torch.randn(cout, generator=g))
        bn_ref.bias.copy_(0.2 * torch.randn(cout, generator=g))
    yr_raw = F.conv2d(img.half().float(), w.half().float(), stride=2, padding=pad)
    yr = F.relu(bn_ref(yr_raw))
    s = hc.ConvSpec(3, cout, k, 2, pad, first=True)
    outs = {}
    for fused in (False, True):
        hc.FUSE_FIRST_LAYER = fused
        bn = nn.BatchNorm2d(cout)
        bn.load_state_dict({k2: v.clone() for k2, v in nn.BatchNorm2d(cout).state_dict().items()})
        with torch.no_grad():
            bn.weight.copy_(bn_ref.weight); bn.bias.copy_(bn_ref.bias)
        bn.to(_dev()).train()
        for p_ in bn.parameters():
            p_.requires_grad = False
        wd = w.to(_dev())
        with runtime.precision(torch.float16, 1.0):
            x = hops.image_to_s2d(img.to(_dev()), torch.float16)
            cfg = hc._Cfg(s, hc.PackedWeights(), bn, True, 1.0)
            y = hc.conv_bn_act(x, wd, None, bn, True, None, cfg, True)
            launched_kernel("conv_s2d_k", what=f"first layer {spec} fused={fused}")
        hc.flush_bn_counters()
        outs[fused] = (y, bn)
    hc.FUSE_FIRST_LAYER = True
    for fused in (False, True):
        y, bn = outs[fused]
        assert rel_err(nchw(y, cout), yr) < 3e-3, fused
        close(bn.running_mean, bn_ref.running_mean, rtol=2e-3, atol=2e-4)
        close(bn.running_var, bn_ref.running_var, rtol=2e-3, atol=2e-4)
        assert int(bn.num_batches_tracked) == 1
    assert rel_err(outs[True][0], outs[False][0]) < 2e-3
    # mode 0 alone: raw conv output and its statistics
    with runtime.precision(torch.float16, 1.0):
        x = hops.image_to_s2d(img.to(_dev()), torch.float16)
        cfg = hc._Cfg(s, hc.PackedWeights(), None, False, 1.0)
        raw = hc.conv_bn_act(x, w.to(_dev()), None, None, False, None, cfg, False)
    assert rel_err(nchw(raw, cout), yr_raw) < 2e-3


@pytest.mark.parametrize("case", [(128, 128, 3, 1, 1, 2, 186, 181), (256, 256, 3, 1, 1, 8, 96, 100), (128, 256, 3, 2, 1, 2, 370, 361)])
def test_conv_big_tile_variant_vs_torch(case):
    """Problems with M >= 65536 rows take the 8-wave ring kernels -- N = 128: conv_ring_k<256,128> (32x32x16 MFMA), N >= 256:
    conv_ring16_k<256,256> (16x16x32 MFMA, transposed product) -- forward and statistics against torch; the launched kernel is
    asserted, so a moved dispatch threshold cannot silently take the test away from them."""
    from self_driving_model_amd.hip import conv as hc
    cin, cout, k, st, pad, B, H, W = case
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).half().float()
    yr = F.conv2d(x, w, stride=st, padding=pad)
    s = hc.ConvSpec(cin, cout, k, st, pad)
    OH, OW = yr.shape[2], yr.shape[3]
    assert B * OH * OW >= 65536
    y = torch.zeros(B, OH, OW, cout, dtype=torch.float16, device=_dev())
    stats = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=_dev())
    from self_driving_model_amd.hip import lib
    old_halo = lib.get().am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, 1 << 30)  # (the 3x3 / stride-1 N = 128 case would take conv_halo_k: its own test)
    try:
        hc.conv_gemm(hc.fwd_geom(s, B, H, W, cin, cout, 2), nhwc(x, torch.float16), hc.pack_fwd(w.to(_dev()), s, torch.float16), None, False, y, stats)
    finally:
        lib.get().am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, old_halo)
    launched_kernel("conv_ring_k<256,128>" if cout == 128 else "conv_ring16_k<256,256>", what=f"big tile {case}")
    torch.cuda.synchronize()
    close(nchw(y, cout), yr, rtol=3e-3, atol=3e-3)
    st_ = stats.view(16, 2, cout).sum(0).cpu()
    np.testing.assert_allclose(st_[0].numpy(), yr.double().sum(dim=(0, 2, 3)).numpy(), rtol=1e-3, atol=1.0)
    np.testing.assert_allclose(st_[1].numpy(), (yr.double() ** 2).sum(dim=(0, 2, 3)).numpy(), rtol=1e-3)


@pytest.mark.parametrize("case", [(512, 512, 3, 1, 1, 16, 23, 40), (512, 256, 3, 1, 1, 32, 23, 40), (256, 512, 3, 2, 1, 16, 45, 80)])
def test_conv_ring16_m128_tile_vs_torch_and_the_tile_it_replaces(case):
    """Problems with N >= 256 that make 58-199 tiles of 256x256 (ResNet layer 4 at B = 16, the 512 -> 256 heads at B = 32): half the
    CUs would idle, so conv_ring16_k runs them on its 128x256 tile (AM_TUNE_RING16_M128_MIN_TILES).  Forward, bias + ReLU epilogue and
    BatchNorm statistics against torch on the f16-rounded operands and against conv_ring_k<256,128>, which took them before."""
    from self_driving_model_amd.hip import conv as hc, lib
    cin, cout, k, st, pad, B, H, W = case
    g = torch.Generator().manual_seed(cin + cout + B)
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).half().float()
    bias = torch.randn(cout, generator=g)
    yr = F.conv2d(x, w, stride=st, padding=pad)
    s = hc.ConvSpec(cin, cout, k, st, pad)
    OH, OW = yr.shape[2], yr.shape[3]
    m128 = -(-B * OH * OW // 128) * -(-cout // 256)
    assert -(-B * OH * OW // 256) * -(-cout // 256) < 200 <= m128
    geom, xd, wp = hc.fwd_geom(s, B, H, W, cin, cout, 2), nhwc(x, torch.float16), hc.pack_fwd(w.to(_dev()), s, torch.float16)
    L = lib.get()
    outs = {}
    old = L.am_get_tuning(lib.AM_TUNE_RING16_M128_MIN_TILES)
    try:
        for gate, name in ((200, "conv_ring16_k<128,256>"), (1 << 30, "conv_ring_k<256,128>")):
            L.am_set_tuning(lib.AM_TUNE_RING16_M128_MIN_TILES, gate)
            y = torch.zeros(B, OH, OW, cout, dtype=torch.float16, device=_dev())
            stats = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=_dev())
            hc.conv_gemm(geom, xd, wp, None, False, y, stats)
            launched_kernel(name, what=f"m128 tile {case} gate={gate}")
            yb = torch.zeros_like(y)
            hc.conv_gemm(geom, xd, wp, bias.to(_dev()), True, yb, None)
            launched_kernel(name, what=f"m128 tile {case} gate={gate} bias+relu")
            torch.cuda.synchronize()
            outs[gate] = (y.float().cpu(), stats.view(16, 2, cout).sum(0).cpu(), yb.float().cpu())
    finally:
        L.am_set_tuning(lib.AM_TUNE_RING16_M128_MIN_TILES, old)
    y, st_, yb = outs[200]
    close(nchw(y, cout), yr, rtol=3e-3, atol=3e-3)
    close(nchw(yb, cout), torch.relu(yr + bias.view(1, -1, 1, 1)), rtol=3e-3, atol=3e-3)
    np.testing.assert_allclose(st_[0].numpy(), yr.double().sum(dim=(0, 2, 3)).numpy(), rtol=1e-3, atol=1.0)
    np.testing.assert_allclose(st_[1].numpy(), (yr.double() ** 2).sum(dim=(0, 2, 3)).numpy(), rtol=1e-3)
    assert rel_err(y, outs[1 << 30][0]) < 1e-3  # (f16 roundings of two fp32 summation orders)
    np.testing.assert_allclose(st_.numpy(), outs[1 << 30][1].numpy(), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("norm", [False, True])
def test_uint8_frames_boundary_matches_reference_preprocessing(norm):
    """uint8 frames -> /255 (-> ImageNet normalise) -> space-to-depth NHWC in one kernel,
    against the reference loader's arithmetic (`read_image(...).float() / 255.0`, torchvision Normalize = sub mean, div std)
    done in torch on the CPU and fed through the fp32 boundary: bit-exact in fp32, identical after rounding in f16; an odd
    frame size takes the fallback path and must agree too."""
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import ops as hops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    try:
        runtime.set_input_normalization(runtime.IMAGENET_MEAN, runtime.IMAGENET_STD) if norm else runtime.set_input_normalization(None)
        for shape in ((2, 3, 64, 96), (1, 3, 33, 47)):
            u8 = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
            ref = u8.float() / 255.0
            if norm:
                ref = (ref - torch.tensor(runtime.IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(runtime.IMAGENET_STD).view(1, 3, 1, 1)
            for dt in (torch.float32, torch.float16):
                a = hops.image_to_s2d(u8.to(dev), dt)
                b = hops.image_to_s2d(ref.to(dev), dt)
                assert a.orig_hw == b.orig_hw == tuple(shape[2:])
                assert torch.equal(a, b), (shape, dt)
    finally:
        runtime.set_input_normalization(None)


@pytest.mark.parametrize("spec", [(64, 7, 3, False), (32, 5, 2, True)])
def test_first_layer_weight_gradient_kernels_vs_torch(spec):
    """Trainable first layer (conv [+bias] -> BatchNorm2d(train) -> ReLU on the image, trajectory_head.py:10-12 / ResNet stem)
    in f16 at a size with ragged edge tiles: the space-to-depth weight-gradient kernel with the BatchNorm backward fused into
    its tile load (am_conv_wgrad_bn) against torch autograd on the same f16-rounded operands, and against the generic
    two-step path (AM_WGRAD_S2D=0 cannot be flipped in-process, so the generic kernel is called directly)."""
    import ctypes
    import torch.nn as nn
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import ops as hops
    cout, k, pad, bias = spec
    B, H, W = 2, 150, 214  # out 75 x 107: partial tiles in both directions
    g = torch.Generator().manual_seed(10 + k)
    img = torch.randn(B, 3, H, W, generator=g)
    w = (torch.randn(cout, 3, k, k, generator=g) / (3 * k * k) ** 0.5)
    bvec = 0.1 * torch.randn(cout, generator=g) if bias else None
    gamma, beta = 1 + 0.2 * torch.randn(cout, generator=g), 0.2 * torch.randn(cout, generator=g)
    probe = torch.randn(B, cout, (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1, generator=g)
    # torch reference on the f16-rounded image / weight
    wr = w.half().float().requires_grad_()
    br = bvec.clone().requires_grad_() if bias else None
    bn_ref = nn.BatchNorm2d(cout).train()
    with torch.no_grad():
        bn_ref.weight.copy_(gamma); bn_ref.bias.copy_(beta)
    (F.relu(bn_ref(F.conv2d(img.half().float(), wr, br, stride=2, padding=pad))) * probe).sum().backward()
    # HIP: the ReLU mask from the sign of the normalised conv output (am_conv_wgrad_bn_sign, am_bn_bwd_reduce_sign: the default)
    # and from the stored activation (am_conv_wgrad_bn, am_bn_bwd_reduce)
    s = hc.ConvSpec(3, cout, k, 2, pad, first=True)
    grads = {}
    for sign in (True, False):
        bn = nn.BatchNorm2d(cout)
        with torch.no_grad():
            bn.weight.copy_(gamma); bn.bias.copy_(beta)
        bn.to(_dev()).train()
        wd = w.to(_dev()).requires_grad_()
        bd = bvec.to(_dev()).requires_grad_() if bias else None
        hc.SIGN_RELU_MASK = sign
        try:
            with runtime.precision(torch.float16, 1.0):
                x = hops.image_to_s2d(img.to(_dev()), torch.float16)
                y = hc.conv_bn_act(x, wd, bd, bn, True, None, hc._Cfg(s, hc.PackedWeights(), bn, True, 1.0), True)
                yf = hops.NhwcToNchw.apply(y, cout, 1.0)
                (yf * probe.to(_dev())).sum().backward()
        finally:
            hc.SIGN_RELU_MASK = True
        assert rel_err(wd.grad, wr.grad) < 2e-2, rel_err(wd.grad, wr.grad)
        assert rel_err(bn.weight.grad, bn_ref.weight.grad) < 1e-2 and rel_err(bn.bias.grad, bn_ref.bias.grad) < 1e-2
        if bias:
            assert float(bd.grad.abs().max()) == 0.0  # a bias in front of a train-mode BN has an exactly zero gradient
        grads[sign] = (wd.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
    for a, b in zip(grads[True], grads[False]):
        assert rel_err(a, b) < 1e-4  # the two masks differ only where the activation underflowed f16
    # the plain (no BatchNorm) form of the same kernel: dW from an explicit conv-output gradient vs torch.nn.grad.conv2d_weight
    with runtime.precision(torch.float16, 1.0):
        x = hops.image_to_s2d(img.to(_dev()), torch.float16)
        geo = hc.fwd_geom(s, B, x.shape[1], x.shape[2], 16, hc.channel_ld(cout, 2), 2, orig_hw=(H, W))
        dz = torch.zeros(B, geo.OH, geo.OW, geo.ldo, dtype=torch.float16)
        dz[..., :cout] = torch.randn(B, geo.OH, geo.OW, cout, generator=torch.Generator().manual_seed(3)).half()
        a = torch.zeros(cout, geo.ntaps * geo.krun, device=_dev())
        hc.conv_wgrad(geo, x, dz.to(_dev()), 1.0, a)
        launched_kernel("conv_s2d_wgrad_k", what=f"first-layer wgrad {spec}")
        dw = hc.unpack_wgrad(a, s, torch.float16)
    ref = torch.nn.grad.conv2d_weight(img.half().float(), w.shape, dz[..., :cout].float().permute(0, 3, 1, 2).contiguous(), stride=2, padding=pad)
    assert rel_err(dw, ref) < 2e-3, rel_err(dw, ref)


@pytest.mark.parametrize("spec", [(64, 7, 3), (32, 5, 2)])
def test_first_layer_fused_bn_weight_gradient_equals_the_two_pass_sequence(spec):
    """am_conv_wgrad_bn / _sign (BatchNorm backward formed in the weight-gradient kernel's tile load) against am_bn_bwd_apply[_sign]
    followed by am_conv_wgrad on the tensor it writes, on an output size with ragged tiles in both directions: the same rounded
    values enter the same MFMAs, so the two agree to the order of the fp32 atomics (a tile pixel outside the image must contribute
    nothing: the affine part of the BatchNorm backward is not zero there)."""
    import ctypes
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import ops as hops
    cout, k, pad = spec
    B, H, W = 2, 150, 214  # conv output 75 x 107
    g = torch.Generator().manual_seed(20 + k)
    L = hc._L()
    s = hc.ConvSpec(3, cout, k, 2, pad, first=True)
    with runtime.precision(torch.float16, 1.0):
        x = hops.image_to_s2d(torch.randn(B, 3, H, W, generator=g).to(_dev()), torch.float16)
    geo = hc.fwd_geom(s, B, x.shape[1], x.shape[2], 16, hc.channel_ld(cout, 2), 2, orig_hw=(H, W))
    OH, OW, ld = geo.OH, geo.OW, geo.ldo
    P = B * OH * OW
    dy = torch.zeros(B, OH, OW, ld, dtype=torch.float16); dy[..., :cout] = torch.randn(B, OH, OW, cout, generator=g).half()
    raw = torch.zeros(B, OH, OW, ld, dtype=torch.float16); raw[..., :cout] = torch.randn(B, OH, OW, cout, generator=g).half()
    dy, raw = dy.to(_dev()), raw.to(_dev())
    mean, rstd = (0.1 * torch.randn(cout, generator=g)).to(_dev()), (0.5 + torch.rand(cout, generator=g)).to(_dev())
    coef = torch.cat([0.5 + torch.rand(cout, generator=g), 0.05 * torch.randn(cout, generator=g), 0.05 * torch.randn(cout, generator=g)]).to(_dev())
    scale, shift = (0.5 + torch.rand(cout, generator=g)).to(_dev()), (0.2 * torch.randn(cout, generator=g)).to(_dev())
    y = torch.relu(raw.float() * torch.nn.functional.pad(scale, (0, ld - cout)) + torch.nn.functional.pad(shift, (0, ld - cout))).half()
    code, p, st = hc.dt_code(torch.float16), hc.ptr, hc.stream()
    ktot = geo.ntaps * geo.krun
    for sign in (True, False):
        dz = torch.zeros_like(dy)
        if sign:
            L.am_bn_bwd_apply_sign(code, p(dy), ld, p(raw), ld, p(mean), p(rstd), p(coef), p(scale), p(shift), p(dz), ld, P, cout, st)
        else:
            L.am_bn_bwd_apply(code, p(dy), ld, p(y), ld, p(raw), ld, p(mean), p(rstd), p(coef), 1, p(dz), ld, None, 0, P, cout, st)
        two_pass = torch.zeros(cout, ktot, device=_dev())
        hc.conv_wgrad(geo, x, dz, 1.0, two_pass)
        fused = torch.zeros(cout, ktot, device=_dev())
        if sign:
            L.am_conv_wgrad_bn_sign(ctypes.byref(geo), code, p(x), p(dy), p(raw), p(mean), p(rstd), p(coef), p(scale), p(shift), 1.0, p(fused), st)
        else:
            L.am_conv_wgrad_bn(ctypes.byref(geo), code, p(x), p(dy), p(y), p(raw), p(mean), p(rstd), p(coef), 1, 1.0, p(fused), st)
        launched_kernel("conv_s2d_wgrad_k", what=f"fused first-layer wgrad {spec}")
        torch.cuda.synchronize()
        assert rel_err(fused, two_pass) < 1e-5, (sign, rel_err(fused, two_pass))


S2_DGRAD_KERNEL = {((128, 256, 4, 240, 256), True): "conv_ring16_k<256,256>", ((128, 256, 4, 240, 256), False): "conv_ring16_k<256,256>",
                   ((64, 128, 2, 320, 320), True): "conv_ring_k<256,128>", ((64, 128, 2, 320, 320), False): "conv_ring_k<256,128>",
                   ((32, 64, 2, 256, 256), True): "conv_gemm2_k", ((32, 64, 2, 256, 256), False): "conv_gemm2_k"}


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(32, 64, 2, 256, 256), (64, 128, 2, 320, 320), (128, 256, 4, 240, 256)])
def test_fused_stride2_dgrad_vs_parity_class_launches_and_torch(case):
    """3x3 / stride-2 / pad-1 input gradient (EasyBackbone convs 2-4, trajectory_head.py:15-22; ResNet stage entries): the
    one-launch form (2x2 dX block per GEMM row, `osplit` output rows) must agree with the four parity-class launches to the
    last bit of their common arithmetic (same products, fp32 accumulation in a different order) and with torch."""
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    cin, cout, B, H, W = case
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    xr, wr = x.clone().requires_grad_(), w.clone()
    yr = F.conv2d(xr.half().float(), wr.half().float(), None, stride=2, padding=1)
    probe = (torch.randn(yr.shape, generator=g)).half().float()
    (yr * probe).sum().backward()
    spec = hc.ConvSpec(cin, cout, 3, 2, 1)
    grads = {}
    for fused in (True, False):
        hc.FUSE_S2_DGRAD = fused
        try:
            with runtime.precision(torch.float16, 1.0):
                xd = nhwc(x, torch.float16).requires_grad_()
                wd = w.to(_dev())
                cfg = hc._Cfg(spec, hc.PackedWeights(), None, False, 1.0)
                if fused:  # the covered shapes must really take the one-launch path
                    assert hc.dgrad_s2_plan(spec, B, H, W, cin, cout, 2) is not None
                y = hc.conv_bn_act(xd, wd, None, None, False, None, cfg, False)
                (y[..., :cout].float() * probe.permute(0, 2, 3, 1).to(_dev())).sum().backward()
                launched_kernel(S2_DGRAD_KERNEL.get((case, fused)), what=f"s2 dgrad {case} fused={fused}")
            grads[fused] = xd.grad.float().cpu()
        finally:
            hc.FUSE_S2_DGRAD = True
    assert rel_err(grads[True], grads[False]) < 2e-3
    close(grads[True], grads[False], rtol=2e-2, atol=2e-2, what="fused vs parity-class dgrad")
    assert rel_err(grads[True].permute(0, 3, 1, 2)[:, :cin], xr.grad) < 3e-3


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(64, 64, 3, 1, 1, True), (128, 256, 3, 2, 1, False), (3, 64, 7, 2, 3, True), (64, 128, 1, 2, 0, False)])
def test_eval_bn_folding_matches_normalise_pass_and_torch(case):
    """Inference (torch.no_grad, eval-mode BatchNorm, f16): the BatchNorm folded into the conv weights / bias epilogue must
    agree with the conv -> normalise-pass sequence and with torch (conv2d -> batch_norm(eval) -> relu), conv bias included."""
    import torch.nn as nn
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import ops as hops
    cin, cout, k, st, pad, relu = case
    B, H, W = 2, 66, 94
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(cout, generator=g)); bn.bias.copy_(0.2 * torch.randn(cout, generator=g))
        bn.running_mean.copy_(0.3 * torch.randn(cout, generator=g)); bn.running_var.copy_(0.5 + torch.rand(cout, generator=g))
    bn.eval()
    with torch.no_grad():
        yr = F.batch_norm(F.conv2d(x.half().float(), w.half().float(), b, stride=st, padding=pad), bn.running_mean, bn.running_var,
                          bn.weight, bn.bias, False, 0.0, bn.eps)
        yr = F.relu(yr) if relu else yr
    spec = hc.ConvSpec(cin, cout, k, st, pad, first=(cin == 3))
    bnd = nn.BatchNorm2d(cout); bnd.load_state_dict(bn.state_dict()); bnd.to(_dev()).eval()
    outs = {}
    for fold in (True, False):
        hc.FOLD_EVAL_BN = fold
        try:
            with runtime.precision(torch.float16, 1.0), torch.no_grad():
                xd = hops.image_to_nhwc(x.to(_dev()), torch.float16) if cin == 3 else nhwc(x, torch.float16)
                pk = hc.PackedWeights()
                cfg = hc._Cfg(spec, pk, bnd, relu, 1.0)
                y = hc.conv_bn_act(xd, w.to(_dev()), b.to(_dev()), bnd, relu, None, cfg, False)
                assert (pk.fold is not None) == fold
            outs[fold] = nchw(y, cout)
        finally:
            hc.FOLD_EVAL_BN = True
    assert rel_err(outs[True], outs[False]) < 2e-3
    assert rel_err(outs[True], yr) < 3e-3 and rel_err(outs[False], yr) < 3e-3


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(128, 256, 3, 2, 1, 4, 128, 160), (64, 128, 3, 1, 1, 2, 96, 128), (256, 512, 3, 1, 1, 4, 64, 64),
                                  (128, 256, 1, 2, 0, 4, 128, 192), (512, 512, 3, 1, 1, 6, 46, 80)])
def test_wgrad_ring_kernel_vs_torch_and_register_staged_kernel(case):
    """Weight gradients whose contraction is long enough (>= 16384 output pixels, N % 128 == 0, output rows >= 32 pixels) run
    the LDS-DMA ring kernel (conv_wgrad_ring.hip): against torch's conv2d weight gradient and against the register-staged
    kernel on the SAME geometry (am_set_tuning(AM_TUNE_WGRAD_RING, 0)), including partial k-tiles, two n-tiles, stride 2 and
    1x1; both legs assert the kernel they launched."""
    import ctypes
    from self_driving_model_amd.hip import conv as hc
    cin, cout, k, st, pad, B, H, W = case
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)).requires_grad_()
    y = F.conv2d(x, w, None, stride=st, padding=pad)
    dyr = (torch.randn(y.shape, generator=g) * 0.5).half().float()
    (y * dyr).sum().backward()
    spec = hc.ConvSpec(cin, cout, k, st, pad)
    xd, dyd = nhwc(x, torch.float16), nhwc(dyr, torch.float16)
    geom = hc.fwd_geom(spec, B, H, W, cin, cout, 2)
    assert B * geom.OH * geom.OW >= 16384 and geom.OW >= 32
    outs = []
    L = hc._L()
    for ring in (1, 0, 2):  # 1: the ring kernel, 2: its 16x16x32 MFMA form on the 256-channel tile
        dwp = torch.zeros(cout, geom.ntaps * geom.krun, dtype=torch.float32, device=_dev())
        old = L.am_set_tuning(2, ring)  # AM_TUNE_WGRAD_RING: 0 pins the register-staged kernel on the same geometry
        try:
            hc.conv_wgrad(geom, xd, dyd, 1.0, dwp)
        finally:
            L.am_set_tuning(2, old)
        launched_kernel("wgrad_ring_k" if ring else "conv_wgrad_k", what=f"wgrad {'ring' if ring else 'register-staged'} {case}")
        outs.append(hc.unpack_wgrad(dwp, spec, torch.float16).cpu())
    assert rel_err(outs[0], w.grad) < 2e-3
    assert rel_err(outs[0], outs[1]) < 1e-3
    assert rel_err(outs[0], outs[2]) < 1e-5  # the two MFMA forms: same products, fp32 sums in another order


@pytest.mark.parametrize("case", [(256, 256, 3, 1, 1, 8, 96, 100, True), (128, 256, 3, 2, 1, 2, 370, 361, False), (64, 256, 3, 1, 1, 4, 128, 128, False)])
def test_ring_kernel_generations_agree(case):
    """conv_ring16_k (AM_TUNE_RING 1 block issue / 2 spread / 3 by wave age) against conv_ring_k (0) through am_set_tuning: the
    same products summed in fp32 in another order -- outputs equal after f16 rounding up to one ulp on a few elements,
    statistics to 1e-6 -- with the bias + ReLU epilogue and without, ragged M tiles included."""
    from self_driving_model_amd.hip import conv as hc
    cin, cout, k, st, pad, B, H, W, with_bias = case
    L = hc._L()
    g = torch.Generator().manual_seed(cin + W)
    x = nhwc(torch.randn(B, cin, H, W, generator=g), torch.float16)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(_dev())
    bias = torch.randn(cout, generator=g).to(_dev()) if with_bias else None
    s = hc.ConvSpec(cin, cout, k, st, pad)
    geom = hc.fwd_geom(s, B, H, W, cin, cout, 2)
    wp = hc.pack_fwd(w, s, torch.float16)
    old = L.am_get_tuning(0)
    res = {}
    try:
        for t, name in ((0, "conv_ring_k<256,256>"), (1, "conv_ring16_k<256,256>"), (2, "conv_ring16_k<256,256>"), (3, "conv_ring16_k<256,256>")):
            L.am_set_tuning(0, t)
            y = torch.zeros(B, geom.OH, geom.OW, cout, dtype=torch.float16, device=_dev())
            stats = None if with_bias else torch.zeros(16 * 2 * cout, dtype=torch.float64, device=_dev())
            hc.conv_gemm(geom, x, wp, bias, with_bias, y, stats)
            launched_kernel(name, what=f"ring generations {case} tuning {t}")
            torch.cuda.synchronize()
            res[t] = (y.float().cpu(), None if stats is None else stats.view(16, 2, cout).sum(0).cpu())
    finally:
        L.am_set_tuning(0, old)
    for t in (1, 2, 3):
        assert rel_err(res[t][0], res[0][0]) < 2e-4, (t, rel_err(res[t][0], res[0][0]))
        if res[0][1] is not None:
            np.testing.assert_allclose(res[t][1].numpy(), res[0][1].numpy(), rtol=1e-6, atol=1e-3)
    yr = F.conv2d(x[:1].float().cpu().permute(0, 3, 1, 2), w.half().float().cpu(), None if bias is None else bias.cpu(), stride=st, padding=pad)
    yr = F.relu(yr) if with_bias else yr
    assert rel_err(res[1][0][:1].permute(0, 3, 1, 2), yr) < 2e-3


@pytest.mark.parametrize("case", [(64, 3, 45, 70), (64, 2, 64, 96), (64, 1, 8, 32), (64, 5, 19, 33),
                                  (128, 2, 45, 40), (128, 1, 8, 16), (128, 3, 19, 33), (128, 11, 24, 48)])
def test_patch_staged_weight_gradient_vs_torch_and_generic_kernel(case):
    """conv_patch_wgrad_k (C -> C channels, C = 64 / 128, 3x3 / stride 1: dY tile and input patch staged once per tile, the whole
    gradient in registers -- for C = 128 split over three workgroups, one per horizontal tap -- one slab per tile stream) against
    torch's conv2d weight gradient and against the generic kernels on the same operands, in the workspace form and in the
    atomic form, on maps with ragged tiles in both directions (a pixel outside the map contributes nothing; the patch border
    is the zero padding), with fewer tiles than streams and with several tiles per stream (XCD-ordered walk)."""
    import ctypes
    from self_driving_model_amd.hip import conv as hc
    C, B, H, W = case
    g = torch.Generator().manual_seed(B * 1000 + H + C)
    x = torch.randn(B, C, H, W, generator=g).half().float()
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * np.sqrt(C))).requires_grad_()
    dyr = (torch.randn(B, C, H, W, generator=g) * 0.5).half().float()
    (F.conv2d(x, w, None, stride=1, padding=1) * dyr).sum().backward()
    spec = hc.ConvSpec(C, C, 3, 1, 1)
    xd, dyd = nhwc(x, torch.float16), nhwc(dyr, torch.float16, ld=C)
    geom = hc.fwd_geom(spec, B, H, W, C, C, 2)
    L = hc._L()
    wparam = torch.nn.Parameter(torch.zeros(C, C, 3, 3, device=_dev()))
    ntiles = B * ((H + 7) // 8) * ((W + 31) // 32 if C == 64 else (W + 15) // 16)
    res = {}
    old = L.am_get_tuning(6)
    try:
        for name, min_tiles in (("patch", 1), ("generic", 1 << 30)):
            kernels = "conv_patch_wgrad_k" if name == "patch" else ("conv_wgrad_k", "wgrad_ring_k")
            L.am_set_tuning(6, min_tiles)  # AM_TUNE_PATCH_WGRAD_MIN_TILES
            nbytes = ctypes.c_longlong(0)
            L.am_conv_wgrad_workspace_bytes(ctypes.byref(geom), hc.dt_code(torch.float16), ctypes.byref(nbytes))
            if name == "patch":  # one slab per tile stream: 256 streams (C = 64), 80 streams of three workgroups (C = 128)
                assert nbytes.value == min(256 if C == 64 else 80, ntiles) * C * 9 * C * 4
            a = hc.conv_wgrad_oihw(geom, xd, dyd, 0.5, wparam, spec).clone()
            launched_kernel(kernels, what=f"patch wgrad {case}")
            b = hc.conv_wgrad_oihw(geom, xd, dyd, 0.5, wparam, spec).clone()
            if name == "patch":  # (the generic kernel's many pixel chunks may meet in atomics: not reproducible bit for bit)
                assert torch.equal(a, b), "the slab sum must be bitwise reproducible"
            dwp = torch.zeros(C, 9 * C, dtype=torch.float32, device=_dev())
            hc.conv_wgrad(geom, xd, dyd, 0.5, dwp)  # atomic form, packed layout
            launched_kernel(kernels, what=f"patch wgrad atomic {case}")
            res[name] = (a, hc.unpack_wgrad(dwp, spec, torch.float16))
    finally:
        L.am_set_tuning(6, old)
    torch.cuda.synchronize()
    ref = 0.5 * w.grad
    for name, (ws_form, atomic_form) in res.items():
        assert rel_err(ws_form, ref) < 2e-3, (name, rel_err(ws_form, ref))
        assert rel_err(atomic_form, ws_form) < 1e-5, (name, rel_err(atomic_form, ws_form))
    # same f16 products summed in fp32 in a different order
    assert rel_err(res["patch"][0], res["generic"][0]) < 1e-5


@pytest.mark.parametrize("case", [(128, 256, 3, 2, 1, 4, 128, 160, torch.float16, "wgrad_ring_k"), (512, 512, 3, 1, 1, 6, 46, 80, torch.float16, "wgrad_ring_k"),
                                  (64, 64, 3, 1, 1, 2, 96, 128, torch.float16, "conv_wgrad_k"), (256, 14, 1, 1, 0, 3, 23, 40, torch.float16, "conv_wgrad_k"),
                                  (64, 128, 3, 2, 1, 2, 23, 40, torch.float32, "conv_wgrad_k")])
def test_wgrad_workspace_form_vs_torch_and_atomic_form(case):
    """am_conv_wgrad_ws (per-chunk slabs in a caller-owned workspace sized by am_conv_wgrad_workspace_bytes + a summing pass that
    writes the nn.Conv2d layout) against torch's conv2d weight gradient and against the atomic form + re-layout, for the ring
    kernel and the register-staged kernel: same values, bitwise reproducible from run to run, `accumulate` adds onto what is
    there, and the workspace size is what the header says (chunks * N * Ktot * 4 bytes, > 0)."""
    import ctypes
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    cin, cout, k, st, pad, B, H, W, dtype, kernel = case
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(B, cin, H, W, generator=g)
    w = (torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)).requires_grad_()
    if dtype == torch.float16:
        x = x.half().float()
    y = F.conv2d(x, w, None, stride=st, padding=pad)
    dyr = torch.randn(y.shape, generator=g) * 0.5
    if dtype == torch.float16:
        dyr = dyr.half().float()
    (y * dyr).sum().backward()
    spec = hc.ConvSpec(cin, cout, k, st, pad)
    es = 2 if dtype == torch.float16 else 4
    ldo = hc.channel_ld(cout, es)
    xd, dyd = nhwc(x, dtype), nhwc(dyr, dtype, ld=ldo)
    geom = hc.fwd_geom(spec, B, H, W, cin, ldo, es)
    L = hc._L()
    nbytes = ctypes.c_longlong(0)
    L.am_conv_wgrad_workspace_bytes(ctypes.byref(geom), hc.dt_code(dtype), ctypes.byref(nbytes))
    ktot = geom.ntaps * geom.krun
    assert nbytes.value > 0 and nbytes.value % (cout * ktot * 4) == 0
    wparam = torch.nn.Parameter(torch.zeros(cout, cin, k, k, device=_dev()))
    outs = []
    old_cap = L.am_set_tuning(3, 1 << 20)  # AM_TUNE_WGRAD_MAX_SLABS: one slab per pixel chunk, however many
    try:
        L.am_conv_wgrad_workspace_bytes(ctypes.byref(geom), hc.dt_code(dtype), ctypes.byref(nbytes))
        slabs = nbytes.value // (cout * ktot * 4)
        for _ in range(2):
            dw = hc.conv_wgrad_oihw(geom, xd, dyd, 0.5, wparam, spec)  # scale 0.5: the loss-scale division
            launched_kernel(kernel, what=f"wgrad ws {case[:8]}")
            outs.append(dw.clone())
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), "the slab sum must be bitwise reproducible"
        L.am_set_tuning(3, 0)  # ... and the other extreme: every chunk adds atomically into ONE zero-filled slab
        L.am_conv_wgrad_workspace_bytes(ctypes.byref(geom), hc.dt_code(dtype), ctypes.byref(nbytes))
        assert nbytes.value == cout * ktot * 4 and slabs >= 1
        one = hc.conv_wgrad_oihw(geom, xd, dyd, 0.5, wparam, spec)
        assert rel_err(one, outs[0]) < 1e-5, rel_err(one, outs[0])
    finally:
        L.am_set_tuning(3, old_cap)
    tol = 2e-3 if dtype == torch.float16 else 1e-5
    assert rel_err(outs[0], 0.5 * w.grad) < tol, rel_err(outs[0], 0.5 * w.grad)
    if dtype == torch.float32:
        close(outs[0], 0.5 * w.grad, rtol=RT, atol=1e-5, what="fp32 weight gradient")
    # atomic form + re-layout
    dwp = torch.zeros(cout, ktot, dtype=torch.float32, device=_dev())
    hc.conv_wgrad(geom, xd, dyd, 0.5, dwp)
    assert rel_err(hc.unpack_wgrad(dwp, spec, dtype), outs[0]) < 1e-4
    # direct mode: added straight into the parameter's preallocated gradient
    wparam.grad = torch.full((cout, cin, k, k), 2.0, device=_dev())
    runtime.set_direct_grads(True)
    try:
        assert hc.conv_wgrad_oihw(geom, xd, dyd, 0.5, wparam, spec) is None
    finally:
        runtime.set_direct_grads(False)
    # (v + 2) - 2 in fp32 costs an ulp of |v| <= 64; the atomic single-slab mode adds its own summation-order noise
    assert rel_err(wparam.grad - 2.0, outs[0]) < 2e-6, rel_err(wparam.grad - 2.0, outs[0])


@pytest.mark.parametrize("case", [(128, 256, 16, 128, 160), (256, 512, 8, 128, 160)])
def test_shortcut_conv_takes_the_two_workgroup_tile(case):
    """1x1 / stride-2 shortcut convolutions (torchvision BasicBlock.downsample, K = Cin: 4 or 8 K-steps) are all prologue and
    epilogue: AM_TUNE_RING_SHORT_K sends them to conv_ring_k<256,128> (two workgroups per CU) although N >= 256; results equal
    the 256x256 ring tile's bit for bit (same products, same fp32 order per output) and torch's."""
    from self_driving_model_amd.hip import conv as hc
    cin, cout, B, H, W = case
    L = hc._L()
    g = torch.Generator().manual_seed(cin)
    x4 = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).half().float()
    s = hc.ConvSpec(cin, cout, 1, 2, 0)
    geom = hc.fwd_geom(s, B, H, W, cin, cout, 2)
    wp = hc.pack_fwd(w.to(_dev()), s, torch.float16)
    xd = nhwc(x4, torch.float16)
    outs = {}
    old = L.am_get_tuning(4)
    try:
        for sk, name in ((8, "conv_ring_k<256,128>"), (0, "conv_ring16_k<256,256>")):
            L.am_set_tuning(4, sk)
            y = torch.zeros(B, geom.OH, geom.OW, cout, dtype=torch.float16, device=_dev())
            hc.conv_gemm(geom, xd, wp, None, False, y, None)
            launched_kernel(name, what=f"shortcut conv {case} short_k={sk}")
            outs[sk] = y.float().cpu()
    finally:
        L.am_set_tuning(4, old)
    assert rel_err(outs[8], outs[0]) < 1e-4
    assert rel_err(nchw(outs[8], cout), F.conv2d(x4, w, stride=2)) < 2e-3


@pytest.mark.parametrize("cin,cout,B,H,W,kernel", [(64, 64, 2, 180, 320, "conv3x3_c64n64_duo_k"), (128, 128, 4, 90, 160, "conv_ring_k<256,128>"),
                                                   (256, 256, 2, 160, 160, "conv_ring16_k<256,256>")])
def test_conv_bias_relu_residual_epilogue(cin, cout, B, H, W, kernel):
    """Inference form of a ResNet block end (eval-mode BatchNorm folded into weights / bias): y = relu(conv(x, w) + b + identity)
    in the conv epilogue (am_conv_gemm_res), and the bias + ReLU epilogue of am_conv_gemm on the same kernels.  Checked against
    torch, and bit for bit against the two-pass sequence (conv + bias rounded to f16, then add + ReLU rounded to f16)."""
    import ctypes
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import lib
    L = lib.get()
    g = torch.Generator().manual_seed(78)
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).half().float()
    b = torch.randn(cout, generator=g)
    r = torch.randn(B, cout, H, W, generator=g).half().float()
    s = hc.ConvSpec(cin, cout, 3, 1, 1)
    geom = hc.fwd_geom(s, B, H, W, cin, cout, 2)
    xd, wp, bd, rd = nhwc(x, torch.float16), hc.pack_fwd(w.to(_dev()), s, torch.float16), b.to(_dev()), nhwc(r, torch.float16)
    conv = F.conv2d(x, w, b, padding=1)
    # bias + ReLU, no residual
    y0 = torch.zeros(B, H, W, cout, dtype=torch.float16, device=_dev())
    hc.conv_gemm(geom, xd, wp, bd, True, y0, None)
    launched_kernel(kernel, what=f"{cin}->{cout} bias + relu")
    close(nchw(y0, cout), F.relu(conv), rtol=3e-3, atol=3e-3)
    # bias only (the first pass of the two-pass sequence)
    y1 = torch.zeros_like(y0)
    hc.conv_gemm(geom, xd, wp, bd, False, y1, None)
    for relu in (True, False):
        y2 = torch.zeros_like(y0)
        L.am_conv_gemm_res(ctypes.byref(geom), hc.dt_code(torch.float16), hc.ptr(xd), hc.ptr(wp), hc.ptr(bd), hc.ptr(rd), int(relu), hc.ptr(y2),
                           hc.stream())
        launched_kernel(kernel, what=f"{cin}->{cout} residual epilogue")
        torch.cuda.synchronize()
        ref = conv + r
        close(nchw(y2, cout), F.relu(ref) if relu else ref, rtol=3e-3, atol=4e-3)
        two_pass = y1.float() + rd.float()
        two_pass = (F.relu(two_pass) if relu else two_pass).half()
        assert torch.equal(y2, two_pass), f"residual epilogue differs from the two-pass sequence (relu={relu})"
