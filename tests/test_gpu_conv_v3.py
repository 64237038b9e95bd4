"""The LDS-DMA / counted-wait 3x3 loop (conv_v3_kernel: 16x16-pixel x 128-channel tiles, 16-bit storage) through the C ABI
vs torch-CPU fp32 on the same rounded inputs: every epilogue it serves, the BatchNorm prologue, the 1x1 second source,
split-K, partial tiles, and agreement with the 128-pixel loop on the same launch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(x, cpad, dt):
    n, c, h, w = x.shape
    out = torch.zeros(n, h, w, cpad, dtype=dt, device="cuda")
    out[..., :c] = x.permute(0, 2, 3, 1).to("cuda").to(dt)
    return out


def _v3(on):
    from pssr2_amd import _lib as L
    old = L.lib().pssr_set_option(b"IGEMM_V3", int(on))
    assert old >= 0
    return old


V3_CASES = [
    # n, cin, cout, h, w, cin1 (1x1 second source, 0 = none)
    (2, 16, 128, 16, 16, 0),      # one chunk: prologue-only pipeline
    (1, 32, 136, 24, 40, 0),      # partial tiles in both directions, two channel tiles with a ragged second one
    (3, 64, 256, 32, 32, 0),      # 4 chunks, 2 channel tiles, 12 pixel tiles
    (2, 48, 128, 32, 48, 32),     # second source: 2 single-tap stages after 9 three-tap stages
    (1, 16, 128, 16, 16, 16),     # one chunk + one second-source chunk
    (2, 256, 128, 16, 16, 0),     # few tiles, 16 chunks: split-K + finish kernel
    (2, 208, 128, 16, 16, 48),    # split-K with an uneven split (13 chunks) and the second source riding with the last slice
    (1, 1024, 128, 16, 16, 0),    # 64 chunks: the BatchNorm table of 1024 channels behind the tiles, split-K 8
    # 64-channel tiles: 16 rows x 32 columns of pixels, the four waves stacked along the rows
    (2, 64, 64, 32, 64, 0),       # 4 chunks, 4 full tiles per image
    (1, 96, 64, 16, 32, 0),       # one tile, 6 chunks
    (3, 32, 48, 24, 40, 16),      # partial tiles in both directions, ragged channel tile, second source
    (2, 256, 64, 16, 32, 0),      # split-K
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", V3_CASES)
def test_v3_forward(case, dt):
    from pssr2_amd import ops, _lib as L
    n, cin, cout, h, w, cin1 = case
    g = torch.Generator().manual_seed(sum(case))
    code = ops.dtype_code(dt)
    yprev = torch.randn(n, cin, h, w, generator=g)
    scale, shift = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    a = F.relu(yprev.to(dt).float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dt).float()
    ref = F.conv2d(a, wt.to(dt).float(), b, padding=1)
    kw = {}
    if cin1:
        x1 = torch.randn(n, cin1, h, w, generator=g)
        w1 = torch.randn(cout, cin1, 1, 1, generator=g) / cin1 ** 0.5
        ref = ref + F.conv2d(x1.to(dt).float(), w1.to(dt).float())
        kw = dict(x1=_nhwc(x1, cin1, dt), cin1=cin1, w1=ops.pack_conv_weight(w1.cuda(), code))
    xd = _nhwc(yprev, cin + 16, dt)                 # slice of a wider buffer
    pw = ops.pack_conv_weight(wt.cuda(), code)
    outs = []
    for on in (2, 0):
        old = _v3(on)
        try:
            out = torch.full((n, h, w, cout + 8), -5.0, dtype=dt, device="cuda")
            stats = torch.zeros(ops.STAT_STRIPES, 2 * cout, dtype=torch.float64, device="cuda")
            ops.conv2d(xd, cin, pw, out, cout, n=n, h=h, w=w, bias=b.cuda(), pro_scale=scale.cuda(), pro_shift=shift.cuda(),
                       flags=L.FLAG_STATS, stats=stats, out_coff=8, **kw)
            torch.cuda.synchronize()
        finally:
            _v3(old)
        assert (out[..., :8] == -5).all()
        got = out[..., 8:].float().cpu().permute(0, 3, 1, 2)
        tol = 2.5e-2 if dt == torch.bfloat16 else 4e-3
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=tol, atol=tol * ref.abs().max().item())
        s = stats.sum(0).cpu().numpy()
        np.testing.assert_allclose(s[:cout], got.double().sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-3)
        np.testing.assert_allclose(s[cout:], (got.double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-3)
        outs.append(got)
    # same products, f32 accumulation in a different order: the two loops agree to rounding of the stored type
    np.testing.assert_allclose(outs[0].numpy(), outs[1].numpy(), rtol=1.6e-2 if dt == torch.bfloat16 else 2e-3,
                               atol=1e-2 * ref.abs().max().item())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 128, 128, 32, 32), (1, 64, 192, 24, 40), (2, 512, 256, 16, 16), (2, 128, 64, 32, 32), (1, 1024, 64, 16, 64)])
def test_v3_dgrad_mask_stats(case, dt):
    """Input gradient through v3 (GEMM-K = cout, GEMM-N = cin) with the ReLU-mask + BatchNorm-backward statistics epilogue,
    and the plain dgrad with a 1x1 second source (the first conv + residual 1x1 of a block)."""
    from pssr2_amd import ops, _lib as L
    n, cout, cin, h, w = case                   # conv: cin -> cout; dgrad reduces over cout
    g = torch.Generator().manual_seed(sum(case) + 7)
    code = ops.dtype_code(dt)
    yprev = torch.randn(n, cin, h, w, generator=g)
    scale, shift = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    mean, invstd = torch.randn(cin, generator=g) * 0.1, torch.rand(cin, generator=g) + 0.5
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cout * 9) ** 0.5
    dy = torch.randn(n, cout, h, w, generator=g)
    yq = yprev.to(dt).float()
    da = torch.nn.grad.conv2d_input(yprev.shape, wt.to(dt).float(), dy.to(dt).float(), padding=1)
    mask = (yq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
    gref = torch.where(mask, da, torch.zeros_like(da))
    xd = _nhwc(yprev, cin, dt)
    old_mode = _v3(2)
    gd = torch.zeros(n, h, w, cin, dtype=dt, device="cuda")
    st2 = torch.zeros(ops.STAT_STRIPES, 2 * cin, dtype=torch.float64, device="cuda")
    pwd = ops.pack_conv_weight(wt.cuda(), code, mode=1)
    ops.conv2d(_nhwc(dy, cout, dt), cout, pwd, gd, cin, n=n, h=h, w=w,
               epilogue=L.EPI_DGRAD_MASK, flags=L.FLAG_STATS, aux=xd, aux_scale=scale.cuda(), aux_shift=shift.cuda(),
               aux_mean=mean.cuda(), aux_invstd=invstd.cuda(), stats=st2)
    torch.cuda.synchronize()
    got = gd.float().cpu().permute(0, 3, 1, 2)
    tol = 2.5e-2 if dt == torch.bfloat16 else 4e-3
    np.testing.assert_allclose(got.numpy(), gref.numpy(), rtol=tol, atol=tol * gref.abs().max().item())
    xhat = (yq - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    s = st2.sum(0).cpu().numpy()
    np.testing.assert_allclose(s[:cin], got.double().sum((0, 2, 3)).numpy(), rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(s[cin:], (got.double() * xhat.double()).sum((0, 2, 3)).numpy(), rtol=1e-5, atol=4e-3)

    # dgrad of a block input: conv3x3^T(dy) + conv1x1^T(dz)
    dz = torch.randn(n, cout, h, w, generator=g)
    w1 = torch.randn(cout, cin, 1, 1, generator=g) / cout ** 0.5
    ref2 = da + torch.nn.grad.conv2d_input(yprev.shape, w1.to(dt).float(), dz.to(dt).float())
    g2 = torch.zeros(n, h, w, cin, dtype=dt, device="cuda")
    ops.conv2d(_nhwc(dy, cout, dt), cout, pwd, g2, cin, n=n, h=h, w=w, x1=_nhwc(dz, cout, dt), cin1=cout,
               w1=ops.pack_conv_weight(w1.cuda(), code, mode=1))
    torch.cuda.synchronize()
    _v3(old_mode)
    got2 = g2.float().cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got2.numpy(), ref2.numpy(), rtol=tol, atol=tol * ref2.abs().max().item())


def test_options_roundtrip():
    """pssr_set_option / pssr_get_option: the table is the only switch (no launch path reads the environment)."""
    from pssr2_amd import _lib as L
    lib = L.lib()
    assert lib.pssr_get_option(b"IGEMM_V3") in (0, 1, 2)
    old = lib.pssr_set_option(b"IGEMM_V3", 0)
    assert lib.pssr_get_option(b"IGEMM_V3") == 0
    assert lib.pssr_set_option(b"IGEMM_V3", old) == 0
    lib.pssr_last_error.restype = __import__("ctypes").c_char_p
    assert lib.pssr_set_option(b"NO_SUCH_OPTION", 1) < 0 and b"unknown option" in lib.pssr_last_error()
    assert lib.pssr_set_option(b"IGEMM_V3", 7) < 0
