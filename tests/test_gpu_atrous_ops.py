"""Kernels of csrc/atrous.hip (through the C ABI) vs torch-CPU references of the same ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DTS = [torch.float32, torch.bfloat16]


def _nhwc(x, cpad, dt, coff=0):
    n, c, h, w = x.shape
    out = torch.full((n, h, w, cpad), 3.0, dtype=dt, device="cuda")
    out[..., coff:coff + c] = x.permute(0, 2, 3, 1).to("cuda").to(dt)
    return out


def _nchw(t, c, coff=0):
    return t[..., coff:coff + c].float().permute(0, 3, 1, 2).cpu()


def _tol(dt):
    return dict(rtol=1e-5, atol=1e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("dil", [1, 3, 7])
def test_im2col_col2im_are_a_dilated_conv_and_its_gradient(dt, dil):
    """conv3x3(dilation d, padding "same") == 1x1 conv over im2col_dil; its input gradient == col2im_dil of the 1x1 input gradient
    (with the pre-activation BatchNorm + ReLU prologue, the ReLU mask and the BatchNorm-backward statistics)."""
    from pssr2_amd import ops
    g = torch.Generator().manual_seed(dil)
    n, c, co, h, w = 2, 20, 16, 15, 18
    code = ops.dtype_code(dt)
    y = torch.randn(n, c, h, w, generator=g).to(dt).float()
    scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    mean, invstd = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    wt = (torch.randn(co, c, 3, 3, generator=g) / (c * 9) ** 0.5).to(dt).float()
    a = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dt).float()
    ref = F.conv2d(a, wt, padding=dil, dilation=dil)
    cp = ops.pad_to(c, 16)
    yd = _nhwc(y, cp + 16, dt, coff=16)
    col = torch.full((n, h, w, 9 * cp), 5.0, dtype=dt, device="cuda")
    ops.im2col_dil(yd, c, col, cp, n, h, w, dil, code, in_coff=16, scale=scale.cuda(), shift=shift.cuda())
    colr = col.float().view(n, h, w, 9, cp)
    assert float(colr[..., c:].abs().max()) == 0.0                    # padded channels are zero
    out = torch.zeros(n, h, w, co, dtype=dt, device="cuda")
    ops.conv2d(col, 9 * cp, ops.pack_conv_weight(wt.cuda(), code, mode=4), out, co, n=n, h=h, w=w)
    torch.cuda.synchronize()
    np.testing.assert_allclose(_nchw(out, co).numpy(), ref.numpy(), **({"rtol": 2e-4, "atol": 2e-4} if dt == torch.float32 else _tol(dt)))
    # gradient
    dy = torch.randn(n, co, h, w, generator=g).to(dt).float()
    da = torch.nn.grad.conv2d_input(a.shape, wt, dy, padding=dil, dilation=dil)
    mask = (y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
    gref = torch.where(mask, da, torch.zeros_like(da))
    dcol = torch.zeros(n, h, w, 9 * cp, dtype=dt, device="cuda")
    ops.conv2d(_nhwc(dy, co, dt), co, ops.pack_conv_weight(wt.cuda(), code, mode=5), dcol, 9 * cp, n=n, h=h, w=w)
    gd = torch.full((n, h, w, cp + 8), -2.0, dtype=dt, device="cuda")
    stats = torch.zeros(ops.STAT_STRIPES, 2 * c, dtype=torch.float64, device="cuda")
    ops.col2im_dil(dcol, cp, gd, c, n, h, w, dil, code, out_coff=8, y=yd, y_coff=16, scale=scale.cuda(), shift=shift.cuda(), mean=mean.cuda(),
                   invstd=invstd.cuda(), stats=stats)
    torch.cuda.synchronize()
    got = _nchw(gd, c, 8)
    assert (gd[..., :8] == -2).all() and (gd[..., 8 + c:] == -2).all()
    np.testing.assert_allclose(got.numpy(), gref.numpy(), **({"rtol": 3e-4, "atol": 3e-4} if dt == torch.float32 else {"rtol": 3e-2, "atol": 3e-2}))
    xhat = (y - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    s = stats.sum(0).cpu().numpy()
    np.testing.assert_allclose(s[:c], got.double().sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[c:], (got.double() * xhat.double()).sum((0, 2, 3)).numpy(), rtol=1e-5, atol=2e-3)
    # plain gather / fold (no prologue, no mask)
    ops.im2col_dil(yd, c, col, cp, n, h, w, dil, code, in_coff=16)
    ref_col = F.unfold(y, 3, dilation=dil, padding=dil).view(n, c, 9, h, w)
    np.testing.assert_array_equal(col.float().view(n, h, w, 9, cp)[..., :c].permute(0, 4, 3, 1, 2).cpu().numpy(), ref_col.numpy())


@pytest.mark.parametrize("dt", DTS)
def test_channel_stats_sum_relu_mask_affine(dt):
    from pssr2_amd import ops
    g = torch.Generator().manual_seed(1)
    n, c, h, w = 3, 36, 9, 11
    code = ops.dtype_code(dt)
    xs = [torch.randn(n, c, h, w, generator=g).to(dt).float() for _ in range(3)]
    bufs = [_nhwc(x, 48 + 4 * i, dt, coff=4 * i) for i, x in enumerate(xs)]
    stats = torch.zeros(ops.STAT_STRIPES, 2 * c, dtype=torch.float64, device="cuda")
    ops.channel_stats_nhwc(bufs[1], c, n * h * w, stats, code, coff=4)
    s = stats.sum(0).cpu().numpy()
    np.testing.assert_allclose(s[:c], xs[1].double().sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(s[c:], (xs[1].double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-4)
    out = torch.full((n, h, w, 64), 9.0, dtype=dt, device="cuda")
    ops.sum_relu([(b, 4 * i) for i, b in enumerate(bufs)], out, n * h * w, c, code, relu=True, out_coff=12)
    ref = F.relu(sum(xs)).to(dt).float()
    np.testing.assert_allclose(_nchw(out, c, 12).numpy(), ref.numpy(), **_tol(dt))
    assert (out[..., :12] == 9).all() and (out[..., 12 + c:] == 9).all()
    dz = torch.zeros(n, h, w, 48, dtype=dt, device="cuda")
    ops.relu_mask(bufs[0], out, dz, n * h * w, c, code, o_coff=12)
    np.testing.assert_array_equal(_nchw(dz, c).numpy(), torch.where(ref > 0, xs[0], torch.zeros_like(xs[0])).numpy())
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    ops.affine_relu(bufs[2], sc.cuda(), sh.cuda(), out, n * h * w, c, code, coff=8, out_coff=0)
    np.testing.assert_allclose(_nchw(out, c).numpy(), F.relu(xs[2] * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).numpy(), **_tol(dt))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("k,h,w", [(1, 8, 8), (2, 12, 16), (4, 16, 16), (8, 16, 24), (2, 13, 9), (4, 10, 14)])
def test_maxpool_k_and_bilinear_up_with_gradients(dt, k, h, w):
    """F.interpolate(F.max_pool2d(x, k), size=(h, w), mode="bilinear") and its gradient (PSP_Pooling, _blocks.py:87), ties included."""
    from pssr2_amd import ops
    g = torch.Generator().manual_seed(k + h)
    n, c = 2, 12
    code = ops.dtype_code(dt)
    x = (torch.randint(-3, 4, (n, c, h, w), generator=g).float() * 0.5).to(dt).float()        # many exact ties
    xr = x.clone().requires_grad_(True)
    pooled = F.max_pool2d(xr, kernel_size=k)
    up = F.interpolate(pooled, size=(h, w), mode="bilinear")
    gy = torch.randn(n, c, h, w, generator=g).to(dt).float()
    up.backward(gy)
    hs, ws = h // k, w // k
    xd = _nhwc(x, 32, dt, coff=4)
    pd = torch.zeros(n, hs, ws, 16, dtype=dt, device="cuda")
    ops.maxpool_k(xd, pd, n, h, w, c, k, code, in_coff=4)
    np.testing.assert_array_equal(_nchw(pd, c).numpy(), pooled.detach().numpy())
    ud = torch.zeros(n, h, w, 16, dtype=dt, device="cuda")
    ops.bilinear_up(pd, ud, n, hs, ws, h, w, c, code)
    np.testing.assert_allclose(_nchw(ud, c).numpy(), up.detach().to(dt).float().numpy(), **_tol(dt))
    dpd = torch.zeros(n, hs, ws, 16, dtype=dt, device="cuda")
    ops.bilinear_up_bwd(_nhwc(gy, 16, dt), dpd, n, hs, ws, h, w, c, code)
    pr = pooled.detach().clone().requires_grad_(True)
    F.interpolate(pr, size=(h, w), mode="bilinear").backward(gy)
    np.testing.assert_allclose(_nchw(dpd, c).numpy(), pr.grad.numpy(), **({"rtol": 1e-5, "atol": 1e-5} if dt == torch.float32 else {"rtol": 3e-2, "atol": 6e-2}))
    dxd = torch.full((n, h, w, 32), 7.0, dtype=dt, device="cuda")
    dpd_exact = _nhwc(pr.grad, 16, dt)
    ops.maxpool_k_bwd(xd, dpd_exact, dxd, n, h, w, c, k, code, act_coff=4, dx_coff=8)
    ref_dx = torch.autograd.grad(F.max_pool2d(xr, kernel_size=k), xr, pr.grad.to(dt).float())[0]
    np.testing.assert_array_equal(_nchw(dxd, c, 8).numpy(), ref_dx.numpy())
    assert (dxd[..., :8] == 7).all() and (dxd[..., 8 + c:] == 7).all()


def test_input_plain():
    from pssr2_amd import ops
    x = torch.rand(2, 3, 7, 9) * 255
    for dt in DTS:
        out = torch.full((2, 7, 9, 16), 4.0, dtype=dt, device="cuda")
        ops.input_plain(x.cuda(), out, ops.dtype_code(dt))
        np.testing.assert_allclose(_nchw(out, 3).numpy(), (x / 128 - 1).to(dt).float().numpy(), rtol=1e-6, atol=1e-6)
        assert (out[..., 3:] == 0).all()
