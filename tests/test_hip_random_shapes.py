"""Randomised shape sweeps of the fused paths against their unfused sequences (same kernels' plain forms, bn.hip / spatial.hip
passes): odd image sizes, ragged tiles in both directions, small batches, channel counts that are not powers of two.  Every
comparison is fused-vs-unfused on identical f16 operands, so the bounds are those of identical arithmetic (bit level up to the
summation order of statistics / atomics), not of f16 against fp32."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a = a.detach().float().cpu().double()
    b = b.detach().float().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _nhwc(x, ld=None):
    B, C, H, W = x.shape
    ld = ld or C
    out = torch.zeros(B, H, W, ld, dtype=torch.float16)
    out[..., :C] = x.permute(0, 2, 3, 1).half()
    return out.to(_dev())


@pytest.mark.parametrize("seed", range(6))
def test_halo_kernel_forms_on_random_shapes(seed):
    """conv_halo_k (forced by AM_TUNE_HALO_MIN_TILES = 1): plain form against the gather kernels, PRE form (BatchNorm + ReLU in the
    staged patch) against am_bn_apply + plain form (bit level), residual epilogue against the two-pass sequence (bit level)."""
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import lib
    L = lib.get()
    rng = np.random.default_rng(100 + seed)
    cin = int(rng.choice([64, 96, 128, 160, 256]))
    cout = int(rng.choice([72, 96, 128]))
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(9, 70)), int(rng.integers(24, 90))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
    r = torch.randn(B, cout, H, W, generator=g)
    b = torch.randn(cout, generator=g).to(_dev())
    sc = (0.5 + torch.rand(cin, generator=g)).to(_dev())
    sh = (0.3 * torch.randn(cin, generator=g)).to(_dev())
    s = hc.ConvSpec(cin, cout, 3, 1, 1)
    ldo = hc.channel_ld(cout, 2)
    geom = hc.fwd_geom(s, B, H, W, cin, ldo, 2)
    xd, wp = _nhwc(x), hc.pack_fwd(w.to(_dev()), s, torch.float16)
    code, p, st = hc.dt_code(torch.float16), hc.ptr, hc.stream()
    P = B * H * W
    old = L.am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, 1)
    try:
        y_h = torch.zeros(B, H, W, ldo, dtype=torch.float16, device=_dev())
        st_h = torch.zeros(16 * 2 * cout, dtype=torch.float64, device=_dev())
        hc.conv_gemm(geom, xd, wp, None, False, y_h, st_h)
        assert L.am_conv_last_variant() == 16
        # PRE form vs explicit normalise pass + plain form
        if cin <= 256:
            xn = torch.empty_like(xd)
            L.am_bn_apply(code, p(xd), cin, p(sc), p(sh), None, 0, 1, p(xn), cin, P, cin, st)
            y_two = torch.zeros_like(y_h)
            hc.conv_gemm(geom, xn, wp, None, False, y_two, None)
            y_pre = torch.zeros_like(y_h)
            L.am_conv_gemm_prebn(ctypes.byref(geom), code, p(xd), p(sc), p(sh), p(wp), p(y_pre), None, st)
            assert L.am_conv_last_variant() == 16
            torch.cuda.synchronize()
            assert torch.equal(y_pre, y_two), "PRE form differs from normalise pass + plain form"
        if ldo == cout:
            rd = _nhwc(r)
            y_b = torch.zeros_like(y_h)
            hc.conv_gemm(geom, xd, wp, b, False, y_b, None)
            y_r = torch.zeros_like(y_h)
            L.am_conv_gemm_res(ctypes.byref(geom), code, p(xd), p(wp), p(b), p(rd), 1, p(y_r), st)
            torch.cuda.synchronize()
            assert torch.equal(y_r, torch.relu(y_b.float() + rd.float()).half())
    finally:
        L.am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, 1 << 30)
    try:
        y_g = torch.zeros_like(y_h)
        st_g = torch.zeros_like(st_h)
        hc.conv_gemm(geom, xd, wp, None, False, y_g, st_g)
        assert L.am_conv_last_variant() != 16
    finally:
        L.am_set_tuning(lib.AM_TUNE_HALO_MIN_TILES, old)
    torch.cuda.synchronize()
    assert rel_err(y_h, y_g) < 1.5e-3  # different fp32 accumulation order over K, f16 outputs
    a, c = st_h.view(16, 2, cout).sum(0).cpu(), st_g.view(16, 2, cout).sum(0).cpu()
    np.testing.assert_allclose(a.numpy(), c.numpy(), rtol=2e-4, atol=0.05)
    assert float(y_h[..., cout:].abs().sum()) == 0.0


@pytest.mark.parametrize("seed", range(5))
def test_stem_pool_fusions_on_random_shapes(seed):
    """am_bn_relu_maxpool3x3s2_fwd against am_bn_apply + am_maxpool3x3s2_fwd (bit level: values and arg-max codes) and
    am_maxpool3x3s2_bwd_bn against am_maxpool3x3s2_bwd + am_bn_bwd_reduce_sign, on odd map sizes and both dtypes."""
    from self_driving_model_amd.hip import conv as hc
    L = hc._L()
    rng = np.random.default_rng(200 + seed)
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(5, 60)), int(rng.integers(5, 70))
    C = int(rng.choice([32, 64, 128]))
    dtype = torch.float16 if seed % 2 == 0 else torch.float32
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn(B, H, W, C, generator=g).to(dtype).to(_dev())
    sc = ((0.5 + torch.rand(C, generator=g)) * torch.where(torch.rand(C, generator=g) < 0.2, -1.0, 1.0)).to(_dev())  # some negative scales
    sh = (0.3 * torch.randn(C, generator=g)).to(_dev())
    mean, rstd = (0.1 * torch.randn(C, generator=g)).to(_dev()), (0.5 + torch.rand(C, generator=g)).to(_dev())
    code, p, st = hc.dt_code(dtype), hc.ptr, hc.stream()
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    P = B * H * W
    y = torch.empty_like(raw)
    L.am_bn_apply(code, p(raw), C, p(sc), p(sh), None, 0, 1, p(y), C, P, C, st)
    pool_a, arg_a = torch.empty(B, OH, OW, C, dtype=dtype, device=_dev()), torch.empty(B, OH, OW, C, dtype=torch.uint8, device=_dev())
    L.am_maxpool3x3s2_fwd(code, p(y), p(pool_a), p(arg_a), B, H, W, C, st)
    pool_b, arg_b = torch.empty_like(pool_a), torch.empty_like(arg_a)
    L.am_bn_relu_maxpool3x3s2_fwd(code, p(raw), p(sc), p(sh), p(pool_b), p(arg_b), B, H, W, C, st)
    torch.cuda.synchronize()
    assert torch.equal(pool_a, pool_b) and torch.equal(arg_a, arg_b)
    dpool = torch.randn(B, OH, OW, C, generator=g).to(dtype).to(_dev())
    dx_a = torch.empty_like(raw)
    L.am_maxpool3x3s2_bwd(code, p(dpool), p(arg_a), p(dx_a), B, H, W, C, st)
    sums_a = torch.zeros(16 * 2 * C, dtype=torch.float64, device=_dev())
    L.am_bn_bwd_reduce_sign(code, p(dx_a), C, p(raw), C, p(mean), p(rstd), p(sc), p(sh), p(sums_a), P, C, st)
    dx_b = torch.empty_like(raw)
    sums_b = torch.zeros_like(sums_a)
    L.am_maxpool3x3s2_bwd_bn(code, p(dpool), p(arg_a), p(dx_b), B, H, W, C, p(raw), p(mean), p(rstd), p(sc), p(sh), p(sums_b), st)
    torch.cuda.synchronize()
    assert torch.equal(dx_a, dx_b)
    a, c = sums_a.view(16, 2, C).sum(0).cpu().numpy(), sums_b.view(16, 2, C).sum(0).cpu().numpy()
    np.testing.assert_allclose(a, c, rtol=1e-5, atol=1e-3 * max(1.0, float(np.abs(a).max()) * 1e-3))


@pytest.mark.parametrize("seed", range(6))
def test_batchnorm_passes_on_odd_channel_counts(seed):
    """am_bn_apply / am_bn_bwd_apply[_sign] / am_bn_bwd_reduce[_sign] with channel counts whose 16-byte chunks per pixel do not
    divide 256 (the per-iteration path) and ones that do (constants hoisted), against torch on the same operands."""
    from self_driving_model_amd.hip import conv as hc
    L = hc._L()
    rng = np.random.default_rng(300 + seed)
    C = int(rng.choice([24, 40, 64, 96, 136, 256]))
    dtype = torch.float16 if seed % 2 else torch.float32
    P = int(rng.integers(50, 3000))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(P, C, generator=g).to(dtype)
    res = torch.randn(P, C, generator=g).to(dtype)
    dy = torch.randn(P, C, generator=g).to(dtype)
    sc, sh = 0.5 + torch.rand(C, generator=g), 0.3 * torch.randn(C, generator=g)
    mean, rstd = 0.1 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
    coef = torch.cat([0.5 + torch.rand(C, generator=g), 0.05 * torch.randn(C, generator=g), 0.05 * torch.randn(C, generator=g)])
    code, p, st = hc.dt_code(dtype), hc.ptr, hc.stream()
    d = lambda t_: t_.to(_dev())
    xd, rd, dyd, scd, shd, md, rsd, cd = d(x), d(res), d(dy), d(sc), d(sh), d(mean), d(rstd), d(coef)
    y = torch.empty_like(xd)
    L.am_bn_apply(code, p(xd), C, p(scd), p(shd), p(rd), C, 1, p(y), C, P, C, st)
    ref = torch.relu(x.float() * sc + sh + res.float()).to(dtype)
    tol = 2e-3 if dtype == torch.float16 else 1e-6
    assert rel_err(y, ref) < tol
    y1 = torch.empty_like(xd)
    L.am_bn_apply(code, p(xd), C, p(scd), p(shd), None, 0, 1, p(y1), C, P, C, st)
    mask = (x.float() * sc + sh) > 0
    dzm = torch.where(mask, dy.float(), torch.zeros(()))
    xhat = (x.float() - mean) * rstd
    dx_ref = coef[:C] * (dzm - coef[C:2 * C] - xhat * coef[2 * C:])
    for sign in (True, False):
        dx = torch.empty_like(xd)
        sums = torch.zeros(16 * 2 * C, dtype=torch.float64, device=_dev())
        if sign:
            L.am_bn_bwd_apply_sign(code, p(dyd), C, p(xd), C, p(md), p(rsd), p(cd), p(scd), p(shd), p(dx), C, P, C, st)
            L.am_bn_bwd_reduce_sign(code, p(dyd), C, p(xd), C, p(md), p(rsd), p(scd), p(shd), p(sums), P, C, st)
        else:
            L.am_bn_bwd_apply(code, p(dyd), C, p(y1), C, p(xd), C, p(md), p(rsd), p(cd), 1, p(dx), C, None, 0, P, C, st)
            L.am_bn_bwd_reduce(code, p(dyd), C, p(y1), C, p(xd), C, p(md), p(rsd), 1, p(sums), P, C, st)
        torch.cuda.synchronize()
        assert rel_err(dx, dx_ref) < (3e-3 if dtype == torch.float16 else 1e-5), (sign, C)
        sm = sums.view(16, 2, C).sum(0).cpu()
        np.testing.assert_allclose(sm[0].numpy(), dzm.double().sum(0).numpy(), rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(sm[1].numpy(), (dzm.double() * xhat.double()).sum(0).numpy(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("seed", range(4))
def test_lsap_random_sizes_against_scipy(seed):
    """lsap_reg_k / lsap_k against scipy on random sizes, integer-valued costs (many exact ties: scipy's tie rule) and float costs,
    both storage orders, ragged column counts."""
    from scipy.optimize import linear_sum_assignment
    from self_driving_model_amd.hip import matcher as hm
    rng = np.random.default_rng(400 + seed)
    B = int(rng.integers(1, 6))
    nr, nc = int(rng.integers(1, 40)), int(rng.integers(1, 1200))
    ties = seed % 2 == 0
    cost = rng.integers(0, 4, size=(B, nr, nc)).astype(np.float32) if ties else rng.random((B, nr, nc)).astype(np.float32)
    ncols = rng.integers(1, nc + 1, size=B).astype(np.int32)
    for transposed in (False, True):
        c_dev = torch.from_numpy(np.ascontiguousarray(cost.transpose(0, 2, 1)) if transposed else cost).to(_dev())
        rows, cols, count, status = hm.lsap_batched(c_dev, torch.from_numpy(ncols), transposed_storage=transposed)
        torch.cuda.synchronize()
        for b_ in range(B):
            r_ref, c_ref = linear_sum_assignment(cost[b_, :, :ncols[b_]])
            k = int(count[b_])
            assert int(status[b_]) == 0 and k == len(r_ref)
            assert np.array_equal(rows[b_, :k].cpu().numpy(), r_ref) and np.array_equal(cols[b_, :k].cpu().numpy(), c_ref), (seed, transposed, b_)


@pytest.mark.parametrize("seed", range(4))
def test_layer1_epilogue_kernel_on_random_shapes(seed):
    """conv3x3_c64n64_duo_k's inference form (bias / ReLU / residual epilogue) on random map sizes with ragged 8x16 tiles: against
    torch, and the residual form bit for bit against conv + bias (rounded) -> + residual -> ReLU (rounded)."""
    from self_driving_model_amd.hip import conv as hc
    L = hc._L()
    rng = np.random.default_rng(500 + seed)
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(150, 260)), int(rng.integers(160, 330))
    if B * H * W < 65536:
        B += 1
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 64, H, W, generator=g).half().float()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().float()
    b = torch.randn(64, generator=g)
    r = torch.randn(B, 64, H, W, generator=g).half().float()
    s = hc.ConvSpec(64, 64, 3, 1, 1)
    geom = hc.fwd_geom(s, B, H, W, 64, 64, 2)
    xd, wp, bd, rd = _nhwc(x), hc.pack_fwd(w.to(_dev()), s, torch.float16), b.to(_dev()), _nhwc(r)
    code, p, st = hc.dt_code(torch.float16), hc.ptr, hc.stream()
    y_b = torch.zeros(B, H, W, 64, dtype=torch.float16, device=_dev())
    hc.conv_gemm(geom, xd, wp, bd, False, y_b, None)
    assert L.am_conv_last_variant() == 3
    y_br = torch.zeros_like(y_b)
    hc.conv_gemm(geom, xd, wp, bd, True, y_br, None)
    y_r = torch.zeros_like(y_b)
    L.am_conv_gemm_res(ctypes.byref(geom), code, p(xd), p(wp), p(bd), p(rd), 1, p(y_r), st)
    assert L.am_conv_last_variant() == 3
    torch.cuda.synchronize()
    ref = F.conv2d(x, w, b, padding=1)
    assert rel_err(y_b.permute(0, 3, 1, 2), ref) < 1e-3
    assert torch.equal(y_br, torch.relu(y_b.float()).half())
    assert torch.equal(y_r, torch.relu(y_b.float() + rd.float()).half())


@pytest.mark.parametrize("seed", range(4))
def test_first_layer_fused_weight_gradient_on_random_image_sizes(seed):
    """conv_s2d_wgrad_k's fused BatchNorm-backward form against am_bn_bwd_apply_sign + the plain form, for the stem (7x7, 64) and the
    policy first layer (5x5, 32), on random image sizes (ragged 8x32 tiles, odd output sizes)."""
    from self_driving_model_amd import runtime
    from self_driving_model_amd.hip import conv as hc
    from self_driving_model_amd.hip import ops as hops
    L = hc._L()
    rng = np.random.default_rng(600 + seed)
    cout, k, pad = (64, 7, 3) if seed % 2 == 0 else (32, 5, 2)
    B, H, W = int(rng.integers(1, 3)), 2 * int(rng.integers(50, 120)), 2 * int(rng.integers(60, 150))
    g = torch.Generator().manual_seed(seed)
    s = hc.ConvSpec(3, cout, k, 2, pad, first=True)
    with runtime.precision(torch.float16, 1.0):
        x = hops.image_to_s2d(torch.randn(B, 3, H, W, generator=g).to(_dev()), torch.float16)
    geo = hc.fwd_geom(s, B, x.shape[1], x.shape[2], 16, hc.channel_ld(cout, 2), 2, orig_hw=(H, W))
    OH, OW, ld = geo.OH, geo.OW, geo.ldo
    if B * OH * OW < 2048:
        pytest.skip("below the patch kernel's size gate")
    P = B * OH * OW
    dy = torch.zeros(B, OH, OW, ld, dtype=torch.float16); dy[..., :cout] = torch.randn(B, OH, OW, cout, generator=g).half()
    raw = torch.zeros(B, OH, OW, ld, dtype=torch.float16); raw[..., :cout] = torch.randn(B, OH, OW, cout, generator=g).half()
    dy, raw = dy.to(_dev()), raw.to(_dev())
    mean, rstd = (0.1 * torch.randn(cout, generator=g)).to(_dev()), (0.5 + torch.rand(cout, generator=g)).to(_dev())
    coef = torch.cat([0.5 + torch.rand(cout, generator=g), 0.05 * torch.randn(cout, generator=g), 0.05 * torch.randn(cout, generator=g)]).to(_dev())
    scale, shift = (0.5 + torch.rand(cout, generator=g)).to(_dev()), (0.2 * torch.randn(cout, generator=g)).to(_dev())
    code, p, st = hc.dt_code(torch.float16), hc.ptr, hc.stream()
    ktot = geo.ntaps * geo.krun
    dz = torch.zeros_like(dy)
    L.am_bn_bwd_apply_sign(code, p(dy), ld, p(raw), ld, p(mean), p(rstd), p(coef), p(scale), p(shift), p(dz), ld, P, cout, st)
    two_pass = torch.zeros(cout, ktot, device=_dev())
    hc.conv_wgrad(geo, x, dz, 1.0, two_pass)
    assert L.am_conv_last_variant() == 15
    fused = torch.zeros(cout, ktot, device=_dev())
    L.am_conv_wgrad_bn_sign(ctypes.byref(geo), code, p(x), p(dy), p(raw), p(mean), p(rstd), p(coef), p(scale), p(shift), 1.0, p(fused), st)
    torch.cuda.synchronize()
    assert rel_err(fused, two_pass) < 1e-5


@pytest.mark.parametrize("seed", range(4))
def test_fused_dense_loss_on_random_sizes(seed):
    """am_upsample_ce2d_* against BilinearUp -> CrossEntropy2d on random low / high resolutions (non-integer scales, maps narrower
    than one column group, one-row maps), both class counts and dtypes."""
    from self_driving_model_amd.hip import ops as hops
    rng = np.random.default_rng(700 + seed)
    C = 3 if seed % 2 else 19
    dtype = torch.float16 if seed < 2 else torch.float32
    B, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(1, 20))
    H, W = int(h * rng.uniform(1.5, 20)), int(w * rng.uniform(1.5, 20))
    ld = C if (C * (2 if dtype == torch.float16 else 4)) % 16 == 0 else 32
    g = torch.Generator().manual_seed(seed)
    low = (torch.randn(B, h, w, C, generator=g) * 2).to(dtype)
    tgt = torch.randint(0, C, (B, H, W), generator=g)
    tgt[torch.rand(B, H, W, generator=g) < 0.2] = 255
    ls = 16.0 if dtype == torch.float16 else 1.0

    def run(fused):
        lowd = torch.zeros(B, h, w, ld, dtype=dtype, device=_dev())
        lowd[..., :C] = low.to(_dev())
        lowd.requires_grad_()
        if fused:
            loss = hops.UpsampleCrossEntropy.apply(lowd, tgt.to(_dev()), C, H, W, 255, ls)
        else:
            loss = hops.CrossEntropy2d.apply(hops.BilinearUp.apply(lowd, C, H, W, ls), tgt.to(_dev()), 255)
        (loss * 0.7).backward()
        torch.cuda.synchronize()
        return float(loss.detach()), lowd.grad.float()

    lf, gf = run(True)
    lu, gu = run(False)
    assert abs(lf - lu) <= 1e-5 * max(1.0, abs(lu))
    assert rel_err(gf, gu) < (2e-3 if dtype == torch.float16 else 1e-5), (B, h, w, H, W, C)
