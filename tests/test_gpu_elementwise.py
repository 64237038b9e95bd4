"""Element-wise / layout kernels with vectorised fast paths (csrc/elementwise.hip) vs torch on the CPU: bit-exact data movement,
statistics within f32 summation noise."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", [(2, 6, 10, 16, 2, 8, 16), (1, 4, 4, 8, 2, 0, 0), (2, 3, 5, 4, 4, 0, 8), (1, 8, 8, 12, 2, 4, 0)])
def test_pixel_shuffle_and_inverse(case, dt):
    """F.pixel_shuffle(x, r) into a channel slice of a wider NHWC buffer and back (resunet.py:82-84): r = 2 with 8-aligned slices
    takes the 16-byte kernel, everything else the generic one; both are pure permutations (bit-exact)."""
    from pssr2_amd import ops
    n, h, w, c_hi, r, lo_off, hi_off = case
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, c_hi * r * r, h, w, generator=g).to(dt)
    want = F.pixel_shuffle(x.float(), r).to(dt)                              # [n, c_hi, h*r, w*r]
    lo = torch.zeros(n, h, w, lo_off + c_hi * r * r + 8, dtype=dt, device="cuda")
    lo[..., lo_off:lo_off + c_hi * r * r] = x.permute(0, 2, 3, 1).cuda()
    hi = torch.full((n, h * r, w * r, hi_off + c_hi + 8), 7.0, dtype=dt, device="cuda")
    ops.pixel_shuffle(lo, hi, n, h, w, c_hi, r, code, lo_coff=lo_off, hi_coff=hi_off)
    got = hi[..., hi_off:hi_off + c_hi].permute(0, 3, 1, 2).cpu()
    assert torch.equal(got, want)
    assert torch.all(hi[..., :hi_off] == 7.0) and torch.all(hi[..., hi_off + c_hi:] == 7.0)      # neighbours untouched
    back = torch.full_like(lo, 3.0)
    ops.pixel_shuffle(back, hi, n, h, w, c_hi, r, code, lo_coff=lo_off, hi_coff=hi_off, inverse=True)
    assert torch.equal(back[..., lo_off:lo_off + c_hi * r * r], lo[..., lo_off:lo_off + c_hi * r * r])
    assert torch.all(back[..., :lo_off] == 3.0)


@pytest.mark.parametrize("c", [64, 24, 256])
def test_bn_backward_elementwise_kernels(c):
    """relu_bwd_stats + bn_bwd_apply (the BatchNorm / ReLU backward of a ResBlock tail, _blocks.py:39-41): power-of-two channel
    counts take the 8-channel kernels, c = 24 the generic ones."""
    from pssr2_amd import ops
    dt, code = torch.bfloat16, ops.dtype_code(torch.bfloat16)
    g = torch.Generator().manual_seed(c)
    npix = 3 * 20 * 12
    dout, out, y = (torch.randn(npix, c, generator=g).to(dt) for _ in range(3))
    mean, invstd = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    dz = torch.empty(npix, c, dtype=dt, device="cuda")
    stats = torch.zeros(ops.STAT_STRIPES * 2 * c, dtype=torch.float64, device="cuda")
    ops.relu_bwd_stats(dout.cuda(), out.cuda(), y.cuda(), mean.cuda(), invstd.cuda(), dz, stats, npix, c, code)
    want_dz = torch.where(out.float() > 0, dout.float(), torch.zeros(())).to(dt)
    assert torch.equal(dz.cpu(), want_dz)
    st = stats.view(ops.STAT_STRIPES, 2, c).sum(0).cpu()
    gz = want_dz.double()
    np.testing.assert_allclose(st[0].numpy(), gz.sum(0).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(st[1].numpy(), (gz * (y.double() - mean.double()) * invstd.double()).sum(0).numpy(), rtol=1e-4, atol=2e-2)
    a, b, cc = (torch.randn(c, generator=g) for _ in range(3))
    dy = torch.empty(npix, c, dtype=dt, device="cuda")
    ops.bn_bwd_apply(dz, y.cuda(), a.cuda(), b.cuda(), cc.cuda(), dy, npix, c, code)
    want = torch.addcmul(torch.addcmul(cc, b, y.float()), a, want_dz.float())          # a*g + (b*y + c), as the kernel's fma order
    assert (dy.cpu().float() - want).abs().max() <= 2.0 ** -7 * want.abs().max()


def test_copy_f32_batch():
    """Many small device-to-device copies in one launch (16 per kernel): bit-exact, sizes from 1 to beyond one grid stride."""
    from pssr2_amd import ops
    g = torch.Generator().manual_seed(0)
    sizes = [1, 3, 64, 257, 1024, 5000, 70000] * 3                               # 21 pairs -> two launches
    srcs = [torch.randn(s, generator=g).cuda() for s in sizes]
    flat = torch.zeros(sum(sizes) + 8, device="cuda")
    dsts, o = [], 4
    for s in sizes:
        dsts.append(flat[o:o + s]); o += s
    ops.copy_f32_batch(list(zip(dsts, srcs)))
    assert all(torch.equal(d, s) for d, s in zip(dsts, srcs))
    assert (flat[:4] == 0).all() and (flat[-4:] == 0).all()
    with pytest.raises(ValueError):
        ops.copy_f32_batch([(dsts[0], srcs[1])])
    with pytest.raises(ValueError):
        ops.copy_f32_batch([(dsts[0].double(), srcs[0].double())])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 8, 12, 64, 16), (3, 4, 6, 8, 0), (1, 16, 16, 256, 64)])
def test_relu_bwd_stats_with_maxpool_backward_in_the_loader(case, dt):
    """pssr_relu_bwd_stats_pool == pssr_maxpool2_bwd followed by pssr_relu_bwd_stats: dz bit for bit (the pooled + skip gradient is rounded to
    the storage type where the separate kernel stored it), statistics to summation order.  Ties inside a window (ReLU zeros) are what the
    first-maximum rule is about: the activations are post-ReLU, a third of them zero."""
    from pssr2_amd import ops
    n, h, w, c, off = case
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(sum(case))
    cs = off + c + 8
    act = torch.zeros(n, h, w, cs, dtype=dt, device="cuda")
    act[..., off:off + c] = torch.relu(torch.randn(n, h, w, c, generator=g) - 0.4).to(dt).cuda()
    dskip = torch.zeros(n, h, w, cs, dtype=dt, device="cuda")
    dskip[..., off:off + c] = torch.randn(n, h, w, c, generator=g).to(dt).cuda()
    dpool = torch.randn(n, h // 2, w // 2, c, generator=g).to(dt).cuda()
    y = torch.randn(n, h, w, c, generator=g).to(dt).cuda()
    mean, invstd = torch.randn(c, generator=g).cuda(), (torch.rand(c, generator=g) + 0.5).cuda()
    npix = n * h * w
    dout = torch.empty(n, h, w, c, dtype=dt, device="cuda")
    ops.maxpool2_bwd(act, dpool, dskip, dout, n, h, w, c, code, act_coff=off, dskip_coff=off)
    dz0 = torch.empty(n, h, w, c, dtype=dt, device="cuda")
    st0 = torch.zeros(ops.STAT_STRIPES * 2 * c, dtype=torch.float64, device="cuda")
    ops.relu_bwd_stats(dout, act, y, mean, invstd, dz0, st0, npix, c, code, out_coff=off)
    dz1 = torch.full_like(dz0, 9.0)
    st1 = torch.zeros_like(st0)
    assert ops.relu_bwd_stats_fused_ok(code, c, h, w)
    ops.relu_bwd_stats_pool(dpool, dskip, off, act, off, y, mean, invstd, dz1, st1, n, h, w, c, code)
    assert torch.equal(dz1, dz0)
    a, b = (s.view(ops.STAT_STRIPES, 2, c).sum(0).cpu().numpy() for s in (st0, st1))
    np.testing.assert_allclose(b, a, rtol=1e-5, atol=1e-4 * npix ** 0.5)
    # and against torch: max_pool2d's own backward + the skip, then the ReLU mask
    av = act[..., off:off + c].float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    F.max_pool2d(av, 2).backward(dpool.float().cpu().permute(0, 3, 1, 2))
    want = (av.grad.permute(0, 2, 3, 1) + dskip[..., off:off + c].float().cpu()).to(dt)
    want = torch.where(act[..., off:off + c].cpu().float() > 0, want.float(), torch.zeros(())).to(dt)
    assert torch.equal(dz1.cpu(), want)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 4, 6, 128, 0), (1, 8, 8, 32, 16), (2, 2, 4, 1024, 0)])
def test_relu_bwd_stats_with_inverse_pixel_shuffle_in_the_loader(case, dt):
    """pssr_relu_bwd_stats_unshuffle == pssr_pixel_shuffle(inverse) followed by pssr_relu_bwd_stats: dz bit for bit, statistics to
    summation order; the high-resolution gradient is the first c / 4 channels of a wider concat-buffer gradient."""
    from pssr2_amd import ops
    n, h, w, c, out_off = case
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(sum(case))
    dhi = torch.randn(n, 2 * h, 2 * w, c // 4 + 24, generator=g).to(dt).cuda()
    out = torch.zeros(n, h, w, out_off + c, dtype=dt, device="cuda")
    out[..., out_off:] = torch.relu(torch.randn(n, h, w, c, generator=g)).to(dt).cuda()
    y = torch.randn(n, h, w, c, generator=g).to(dt).cuda()
    mean, invstd = torch.randn(c, generator=g).cuda(), (torch.rand(c, generator=g) + 0.5).cuda()
    npix = n * h * w
    dout = torch.empty(n, h, w, c, dtype=dt, device="cuda")
    ops.pixel_shuffle(dout, dhi, n, h, w, c // 4, 2, code, inverse=True)
    dz0 = torch.empty(n, h, w, c, dtype=dt, device="cuda")
    st0 = torch.zeros(ops.STAT_STRIPES * 2 * c, dtype=torch.float64, device="cuda")
    ops.relu_bwd_stats(dout, out, y, mean, invstd, dz0, st0, npix, c, code, out_coff=out_off)
    dz1 = torch.full_like(dz0, 9.0)
    st1 = torch.zeros_like(st0)
    assert ops.relu_bwd_stats_fused_ok(code, c, unshuffle=True)
    ops.relu_bwd_stats_unshuffle(dhi, out, out_off, y, mean, invstd, dz1, st1, n, h, w, c, code)
    assert torch.equal(dz1, dz0)
    a, b = (s.view(ops.STAT_STRIPES, 2, c).sum(0).cpu().numpy() for s in (st0, st1))
    np.testing.assert_allclose(b, a, rtol=1e-5, atol=1e-4 * npix ** 0.5)
    want = F.pixel_unshuffle(dhi[..., :c // 4].float().cpu().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    want = torch.where(out[..., out_off:].cpu().float() > 0, want, torch.zeros(())).to(dt)
    assert torch.equal(dz1.cpu(), want)
