"""HIP implicit-GEMM conv (through the C ABI) vs torch-CPU fp32 F.conv2d on the same inputs."""
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


def _tol(dt, k):
    return (2e-5 * max(1, k) ** 0.5, 1e-5) if dt == torch.float32 else (2.5e-2, 2e-2)


CASES = [
    # n, cin, cout, h, w, ks
    (2, 16, 64, 16, 16, 3),
    (1, 32, 128, 24, 40, 3),     # partial tiles in both directions
    (3, 48, 16, 8, 8, 3),        # GEO 1 (2 images / tile), odd batch
    (9, 16, 32, 4, 4, 3),        # GEO 2
    (33, 16, 8, 2, 2, 3),        # GEO 3
    (130, 32, 36, 1, 1, 3),      # GEO 4
    (2, 64, 256, 16, 16, 1),     # 1x1
    (2, 80, 12, 12, 20, 3),      # cout not multiple of 32
]


@pytest.fixture(params=[2, 0], ids=["v3", "legacy"])
def pipeline_mode(request):
    """Run a test with the LDS-DMA v3 loop wherever the shape allows and with it off (the 128-pixel loop everywhere)."""
    from pssr2_amd import _lib as L
    old3 = L.lib().pssr_set_option(b"IGEMM_V3", request.param)
    assert old3 >= 0
    yield request.param
    L.lib().pssr_set_option(b"IGEMM_V3", old3)


CASES += [
    (2, 48, 128, 32, 48, 3),     # >= 16x16: eligible for the pipelined loop (3 chunks, BN = 128)
    (3, 16, 64, 16, 40, 3),      # BN = 64, partial tiles in x, one chunk
    (2, 80, 192, 24, 24, 1),     # 1x1, odd chunk count (5), two N tiles, partial tiles
    (1, 32, 72, 36, 20, 1),      # 1x1, BN = 128 with cout not a multiple of 32
    (2, 256, 128, 16, 16, 3),    # few tiles, long K: split-K (4 slices) + finish kernel
    (3, 640, 64, 8, 8, 1),       # 1x1 split-K, BN = 64, GEO 1
    (1, 144, 200, 12, 12, 3),    # split-K with an uneven chunk split and partial tiles
    (2, 144, 96, 20, 20, 1),     # 1x1 stage-of-chunks loop: exactly one 9-chunk stage (bf16), partial tiles
    (1, 1040, 64, 16, 16, 1),    # 1x1, 65 chunks: 9-chunk stages with a 2-chunk tail (zero-filled slices), BN = 64
    (2, 208, 40, 8, 8, 1),       # 1x1, 13 chunks: 4-chunk stages with a 1-chunk tail, GEO 1, BN = 64 with cout % 32 != 0
    (1, 2048, 256, 8, 8, 1),     # 1x1 split-K over stages
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv_forward_and_dgrad(case, dt, pipeline_mode):
    from pssr2_amd import ops, _lib as L
    n, cin, cout, h, w, ks = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    b = torch.randn(cout, generator=g)
    code = ops.dtype_code(dt)
    xq = x.to(dt).float()
    wq = wt.to(dt).float()
    ref = F.conv2d(xq, wq, b, padding=ks // 2)

    cpad = ops.pad_to(cin, 16)
    xd = _nhwc(x, cpad + 16, dt)              # slice of a wider buffer: channel offset 16
    xd = torch.cat([torch.full_like(xd[..., :16], 7.0), xd[..., :cpad]], -1).contiguous()
    pw = ops.pack_conv_weight(wt.cuda(), code, mode=0)
    out = torch.full((n, h, w, cout + 8), -5.0, dtype=dt, device="cuda")
    ops.conv2d(xd, cpad, pw, out, cout, n=n, h=h, w=w, in0_coff=16, out_coff=4, bias=b.cuda())
    torch.cuda.synchronize()
    got = out[..., 4:4 + cout].float().cpu().permute(0, 3, 1, 2)
    rtol, atol = _tol(dt, cin * ks * ks)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=rtol, atol=atol * ref.abs().max().item())
    assert (out[..., :4] == -5).all() and (out[..., 4 + cout:] == -5).all()

    # input gradient: mode-1 weights, GEMM-K = cout, GEMM-N = cin
    dy = torch.randn(n, cout, h, w, generator=g)
    dyq = dy.to(dt).float()
    ref_dx = torch.nn.grad.conv2d_input(x.shape, wq, dyq, padding=ks // 2)
    pwd = ops.pack_conv_weight(wt.cuda(), code, mode=1)
    dyd = _nhwc(dy, ops.pad_to(cout, 16), dt)
    cin4 = ops.pad_to(cin, 4)
    dx = torch.zeros(n, h, w, cin4, dtype=dt, device="cuda")
    ops.conv2d(dyd, ops.pad_to(cout, 16), pwd, dx, cin4, n=n, h=h, w=w)
    torch.cuda.synchronize()
    got = dx[..., :cin].float().cpu().permute(0, 3, 1, 2)
    rtol, atol = _tol(dt, cout * ks * ks)
    np.testing.assert_allclose(got.numpy(), ref_dx.numpy(), rtol=rtol, atol=atol * ref_dx.abs().max().item())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_conv_fused_prologue_epilogues(dt, pipeline_mode):
    from pssr2_amd import ops, _lib as L
    g = torch.Generator().manual_seed(3)
    n, cin, cout, h, w = 2, 32, 64, 16, 24
    code = ops.dtype_code(dt)
    yprev = torch.randn(n, cin, h, w, generator=g)
    scale, shift = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    yq = yprev.to(dt).float()
    a = F.relu(yq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dt).float()
    ref = F.conv2d(a, wt.to(dt).float(), b, padding=1)
    refq = ref.to(dt).float()

    xd = _nhwc(yprev, cin, dt)
    pw = ops.pack_conv_weight(wt.cuda(), code)
    out = torch.zeros(n, h, w, cout, dtype=dt, device="cuda")
    stats = torch.zeros(ops.STAT_STRIPES, 2 * cout, dtype=torch.float64, device="cuda")
    ops.conv2d(xd, cin, pw, out, cout, n=n, h=h, w=w, bias=b.cuda(), pro_scale=scale.cuda(), pro_shift=shift.cuda(),
               flags=L.FLAG_STATS, stats=stats)
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    rtol, atol = _tol(dt, cin * 9)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=rtol, atol=atol * ref.abs().max().item())
    # statistics are those of the stored (rounded) values
    s = stats.sum(0).cpu().numpy()      # statistics are striped over PSSR_STAT_STRIPES copies
    np.testing.assert_allclose(s[:cout], got.double().sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(s[cout:], (got.double() ** 2).sum((0, 2, 3)).numpy(), rtol=1e-6, atol=1e-4)

    # tail epilogue: relu(conv1x1(x) + bias + aux*scale + shift)
    w1 = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    y4 = torch.randn(n, cout, h, w, generator=g)
    s4, h4 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    x_in = torch.randn(n, cin, h, w, generator=g)
    ref = F.relu(F.conv2d(x_in.to(dt).float(), w1.to(dt).float(), b) + y4.to(dt).float() * s4.view(1, -1, 1, 1) + h4.view(1, -1, 1, 1))
    out2 = torch.zeros(n, h, w, cout, dtype=dt, device="cuda")
    ops.conv2d(_nhwc(x_in, cin, dt), cin, ops.pack_conv_weight(w1.cuda(), code), out2, cout, n=n, h=h, w=w, bias=b.cuda(),
               epilogue=L.EPI_TAIL, aux=_nhwc(y4, cout, dt), aux_scale=s4.cuda(), aux_shift=h4.cuda())
    torch.cuda.synchronize()
    got = out2.float().cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=rtol, atol=atol * ref.abs().max().item())

    # dgrad + relu mask + BN-backward statistics
    dy = torch.randn(n, cout, h, w, generator=g)
    mean, invstd = torch.randn(cin, generator=g) * 0.1, torch.rand(cin, generator=g) + 0.5
    da = torch.nn.grad.conv2d_input(yprev.shape, wt.to(dt).float(), dy.to(dt).float(), padding=1)
    mask = (yq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
    gref = torch.where(mask, da, torch.zeros_like(da))
    gd = torch.zeros(n, h, w, cin, dtype=dt, device="cuda")
    st2 = torch.zeros(ops.STAT_STRIPES, 2 * cin, dtype=torch.float64, device="cuda")
    ops.conv2d(_nhwc(dy, cout, dt), cout, ops.pack_conv_weight(wt.cuda(), code, mode=1), gd, cin, n=n, h=h, w=w,
               epilogue=L.EPI_DGRAD_MASK, flags=L.FLAG_STATS, aux=xd, aux_scale=scale.cuda(), aux_shift=shift.cuda(),
               aux_mean=mean.cuda(), aux_invstd=invstd.cuda(), stats=st2)
    torch.cuda.synchronize()
    got = gd.float().cpu().permute(0, 3, 1, 2)
    rtol, atol = _tol(dt, cout * 9)
    np.testing.assert_allclose(got.numpy(), gref.numpy(), rtol=rtol, atol=atol * gref.abs().max().item())
    xhat = (yq - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    s = st2.sum(0).cpu().numpy()
    np.testing.assert_allclose(s[:cin], got.double().sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[cin:], (got.double() * xhat.double()).sum((0, 2, 3)).numpy(), rtol=1e-5, atol=2e-3)


def test_conv_two_sources_and_errors(pipeline_mode):
    from pssr2_amd import ops, _lib as L
    g = torch.Generator().manual_seed(5)
    n, h, w, c0, c1, cout = 2, 16, 16, 32, 1, 64
    x0, x1 = torch.randn(n, c0, h, w, generator=g), torch.randn(n, c1, h, w, generator=g)
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) / ((c0 + c1) * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(torch.cat([x0, x1], 1), wt, b, padding=1))
    # source 1 = 3x3 neighbourhood of x1 unrolled into 9 (padded 16) channels, consumed as a flat-K 1x1 conv
    col = F.unfold(x1, 3, padding=1).view(n, c1 * 9, h, w)
    pw0 = ops.pack_conv_weight(wt.cuda(), L.F32, mode=0, ci_begin=0, ci_count=c0)
    pw1 = ops.pack_conv_weight(wt.cuda(), L.F32, mode=2, ci_begin=c0, ci_count=c1)
    out = torch.zeros(n, h, w, cout, device="cuda")
    ops.conv2d(_nhwc(x0, c0, torch.float32), c0, pw0, out, cout, n=n, h=h, w=w, bias=b.cuda(),
               x1=_nhwc(col, 16, torch.float32), cin1=16, w1=pw1, flags=L.FLAG_RELU)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=2e-4, atol=2e-5)
    with pytest.raises(RuntimeError, match="cin0"):
        ops.conv2d(_nhwc(x0, c0, torch.float32), 24 + 1, pw0, out, cout, n=n, h=h, w=w)


WG_CASES = [
    # n, cin, cout, h, w, ks
    (2, 16, 64, 16, 16, 3),
    (1, 32, 128, 24, 40, 3),
    (3, 48, 16, 8, 8, 3),
    (9, 64, 32, 4, 4, 3),
    (2, 64, 256, 16, 16, 1),
    (2, 80, 24, 12, 20, 3),
    (40, 32, 64, 16, 16, 3),     # many pixel tiles per workgroup: exercises the prefetched tile loop
    (4, 128, 128, 32, 32, 3),
    (4, 208, 72, 16, 32, 1),     # 1x1, 128x128 slabs with channel tails on both sides (bf16: lean-loader kernel)
    (4, 160, 96, 8, 8, 1),       # 1x1, 8x8 tiles of 2 images
    (6, 320, 256, 32, 32, 1),    # 1x1, several slabs and pixel tiles per workgroup
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad(case, dt):
    from pssr2_amd import ops
    n, cin, cout, h, w, ks = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 1)
    code = ops.dtype_code(dt)
    yprev = torch.randn(n, cin, h, w, generator=g)
    scale, shift = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    dy = torch.randn(n, cout, h, w, generator=g)
    a = F.relu(yprev.to(dt).float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(a, (cout, cin, ks, ks), dy.to(dt).float(), padding=ks // 2)
    cpad = ops.pad_to(cin, 16)
    copad = ops.pad_to(cout, 16)
    sc = torch.zeros(cpad); sc[:cin] = scale
    sh = torch.zeros(cpad); sh[:cin] = shift
    co_eff = cout if (cout * (2 if dt == torch.bfloat16 else 4)) % 16 == 0 else copad
    dwp = torch.zeros(co_eff, ks * ks, cpad, device="cuda")
    ops.conv2d_wgrad(_nhwc(dy, copad, dt), co_eff, _nhwc(yprev, cpad, dt), cpad, ks * ks, dwp, n=n, h=h, w=w, dtype=code,
                     pro_scale=sc.cuda(), pro_shift=sh.cuda())
    dw = torch.full((cout, cin, ks, ks), 9.0, device="cuda")
    ops.unpack_conv_wgrad(dwp, dw, k_pad=cpad)
    torch.cuda.synchronize()
    tol = 1e-4 if dt == torch.float32 else 2e-2
    np.testing.assert_allclose(dw.cpu().numpy(), ref.numpy(), rtol=tol, atol=tol * ref.abs().max().item())
    # partial-slab mode (no atomics, no zero-initialised destination): parts summed by the unpack pass
    parts = ops.conv2d_wgrad_parts(_nhwc(dy, copad, dt), co_eff, _nhwc(yprev, cpad, dt), cpad, ks * ks, n=n, h=h, w=w, dtype=code,
                                   pro_scale=sc.cuda(), pro_shift=sh.cuda())
    assert parts.dim() == 4 and parts.shape[1:] == (co_eff, ks * ks, cpad)
    dw2 = torch.full((cout, cin, ks, ks), 9.0, device="cuda")
    ops.unpack_conv_wgrad(parts, dw2, k_pad=cpad)
    np.testing.assert_allclose(dw2.cpu().numpy(), ref.numpy(), rtol=tol, atol=tol * ref.abs().max().item())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_conv2d_dual_source_split_k(dt):
    """Few tiles, long K, two sources (the decoder's first conv + residual 1x1 on small maps): K is split over blockIdx.y and the
    1x1 source rides with the last slice; odd chunk counts leave some slices short or empty."""
    from pssr2_amd import ops, _lib as L
    code = ops.dtype_code(dt)
    for n, cin, cin1, cout, h, w in [(2, 416, 96, 128, 8, 8), (1, 1056, 64, 256, 8, 8), (2, 288, 160, 64, 16, 8)]:
        g = torch.Generator().manual_seed(cin + cin1)
        x = torch.randn(n, cin, h, w, generator=g).to(dt).float()
        x1 = torch.randn(n, cin1, h, w, generator=g).to(dt).float()
        w3 = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dt).float()
        w1 = (torch.randn(cout, cin1, 1, 1, generator=g) / cin1 ** 0.5).to(dt).float()
        b = torch.randn(cout, generator=g)
        ref = F.relu(F.conv2d(x, w3, b, padding=1) + F.conv2d(x1, w1))
        out = torch.empty(n, h, w, cout, dtype=dt, device="cuda")
        pw3 = ops.pack_conv_weight(w3.cuda(), code)
        pw1 = ops.pack_conv_weight(w1.cuda(), code)
        ops.conv2d(_nhwc(x, cin, dt), cin, pw3, out, cout, n=n, h=h, w=w, bias=b.cuda(), x1=_nhwc(x1, cin1, dt), cin1=cin1, w1=pw1,
                   flags=L.FLAG_RELU)
        tol = 2e-2 if dt == torch.bfloat16 else 1e-4
        torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=tol, atol=tol * float(ref.abs().max()))


WGD_CASES = [
    # n, cin, cout, h, w: 3x3, 64x64 slabs (what conv_wgrad16d_kernel serves)
    (4, 64, 64, 32, 32),       # full channel tiles, 8x16 pixel tiles, several tiles per workgroup
    (2, 96, 128, 16, 48),      # channel tail on the input side (96 = 64 + 32), two cout tiles
    (40, 64, 64, 16, 16),      # many pixel tiles per workgroup: both LDS buffers in rotation, odd tile counts
    (8, 64, 64, 8, 8),         # 8x8 maps: tiles of two images (GEO 1), every halo pixel on an image border
    (3, 128, 80, 16, 16),      # cout tail (80 = 64 + 16)
    (1, 64, 64, 8, 16),        # a single pixel tile: no prefetch at all
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", WGD_CASES)
def test_conv_wgrad_all_dma_kernel(case, dt):
    """Prologue-free 3x3 weight gradient through the all-DMA kernel (both operands global -> LDS by buffer_load ... lds, double-buffered;
    tunable WGRAD_DMA): against torch's conv2d_weight on the same 16-bit operands, and bit for bit against the register-staged kernel
    (same tiles in the same order, same MFMA sequence: identical partial slabs)."""
    from pssr2_amd import ops, _lib as L
    n, cin, cout, h, w = case
    g = torch.Generator().manual_seed(sum(case))
    code = ops.dtype_code(dt)
    a = torch.randn(n, cin, h, w, generator=g).to(dt).float()
    dy = torch.randn(n, cout, h, w, generator=g).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(a, (cout, cin, 3, 3), dy, padding=1)
    cpad, copad = ops.pad_to(cin, 16), ops.pad_to(cout, 16)

    def run(dma):
        old = L.lib().pssr_set_option(b"WGRAD_DMA", dma)
        try:
            parts = ops.conv2d_wgrad_parts(_nhwc(dy, copad, dt), cout, _nhwc(a, cpad, dt), cpad, 9, n=n, h=h, w=w, dtype=code)
            dw = torch.full((cout, cin, 3, 3), 9.0, device="cuda")
            ops.unpack_conv_wgrad(parts, dw, k_pad=cpad)
            torch.cuda.synchronize()
            return parts.clone(), dw
        finally:
            L.lib().pssr_set_option(b"WGRAD_DMA", old)
    p1, dw1 = run(1)
    p0, dw0 = run(0)
    np.testing.assert_allclose(dw1.cpu().numpy(), ref.numpy(), rtol=2e-2, atol=2e-2 * ref.abs().max().item())
    assert torch.equal(p1, p0) and torch.equal(dw1, dw0)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_bn_relu_apply_equals_the_loader_prologue(dt):
    """pssr_bn_relu_apply writes relu(scale * y + shift) exactly as the convolution loaders' BatchNorm+ReLU prologue stages it: the
    weight gradient with the prologue on the raw tensor == the prologue-free weight gradient on the materialised tensor, bit for bit."""
    from pssr2_amd import ops
    n, c, cout, h, w = 4, 64, 64, 16, 32
    g = torch.Generator().manual_seed(5)
    code = ops.dtype_code(dt)
    y = _nhwc(torch.randn(n, c, h, w, generator=g) * 3, c, dt)
    dy = _nhwc(torch.randn(n, cout, h, w, generator=g), cout, dt)
    scale, shift = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    act = torch.zeros_like(y)
    ops.bn_relu_apply(y, scale, shift, act, n * h * w, c, code)
    want = torch.relu(y.float() * scale + shift)           # (torch rounds the product; the kernel's FMA does not: compare within an ulp)
    torch.testing.assert_close(act.float(), want, rtol=2e-3 if dt == torch.float16 else 1e-2, atol=1e-3)
    assert torch.equal(act > 0, want.to(dt) > 0) or float(((act > 0) != (want.to(dt) > 0)).float().mean()) < 1e-4
    with_pro = ops.conv2d_wgrad_parts(dy, cout, y, c, 9, n=n, h=h, w=w, dtype=code, pro_scale=scale, pro_shift=shift)
    without = ops.conv2d_wgrad_parts(dy, cout, act, c, 9, n=n, h=h, w=w, dtype=code)
    assert torch.equal(with_pro, without)
