"""Launch-to-launch identity of the statistic sums ("Run-to-run"): the f64 statistic rows a convolution leaves, and the
per-image channel sums of RDNet's gate, are the same bits on every launch -- at the full c2 / c3 layer shapes, where thousands of
workgroups add into the same rows in whatever order they finish."""
import pytest
import torch

pytestmark = pytest.mark.gpu
REPS = 8


def _rows(fn, co):
    from pssr2_amd import ops
    first = None
    for _ in range(REPS):
        stats = torch.zeros(ops.STAT_STRIPES * 2 * co, dtype=torch.float64, device="cuda")
        out = fn(stats)
        torch.cuda.synchronize()
        if first is None:
            first = (stats.clone(), out.clone())
        else:
            assert torch.equal(out, first[1]), "outputs differ between launches"
            assert torch.equal(stats, first[0]), f"statistic rows differ between launches (max {float((stats - first[0]).abs().max()):.3e})"
    assert float(first[0].abs().max()) > 0


@pytest.mark.parametrize("h,ci,co", [(128, 64, 64), (64, 128, 128), (32, 256, 256)])
@pytest.mark.parametrize("kind", ["forward", "dgrad"])
def test_conv3x3_statistics_identical_on_every_launch(h, ci, co, kind):
    from pssr2_amd import _lib as L, ops
    n, dt, code = 32, torch.bfloat16, L.BF16
    torch.manual_seed(h + ci)
    x = torch.randn(n, h, h, ci, device="cuda").to(dt)
    aux = torch.randn(n, h, h, co, device="cuda").to(dt)
    pw = ops.pack_conv_weight(torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5, code)
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    a1, a2, a3, a4 = (torch.rand(co, device="cuda") + 0.5 for _ in range(4))
    bias = torch.randn(co, device="cuda")
    if kind == "forward":       # BatchNorm prologue + bias + statistics (Engine._block_forward)
        kw = dict(bias=bias, pro_scale=sc, pro_shift=sh, flags=L.FLAG_STATS)
    else:                       # ReLU mask + statistics of the data gradient (Engine._block_backward)
        kw = dict(epilogue=L.EPI_DGRAD_MASK, flags=L.FLAG_STATS, aux=aux, aux_scale=a1, aux_shift=a2, aux_mean=a3, aux_invstd=a4)

    def launch(stats):
        out = torch.zeros(n, h, h, co, device="cuda", dtype=dt)
        ops.conv2d(x, ci, pw, out, co, n=n, h=h, w=h, stats=stats, **kw)
        return out
    _rows(launch, co)


@pytest.mark.parametrize("h,ci,co", [(128, 32, 128), (64, 64, 256)])
def test_conv1x1_gelu_derivative_statistics_identical_on_every_launch(h, ci, co):
    """RDNet's second 1x1 convolution, data gradient with the GELU-derivative epilogue + statistics (RDEngine: the bias gradient of the
    first 1x1 convolution comes out of these sums)."""
    from pssr2_amd import _lib as L, ops
    n, dt, code = 32, torch.bfloat16, L.BF16
    torch.manual_seed(h)
    x = (torch.randn(n, h, h, ci, device="cuda") * 1e-3).to(dt)
    z = torch.randn(n, h, h, co, device="cuda").to(dt)
    pw = ops.pack_conv_weight(torch.randn(ci, co, 1, 1, device="cuda") / co ** 0.5, code, mode=1)

    def launch(stats):
        out = torch.zeros(n, h, h, co, device="cuda", dtype=dt)
        ops.conv2d(x, ci, pw, out, co, n=n, h=h, w=h, epilogue=L.EPI_DGRAD_GELU, flags=L.FLAG_STATS, aux=z, stats=stats)
        return out
    _rows(launch, co)


def test_image_channel_dot_identical_on_every_launch_and_right():
    from pssr2_amd import _lib as L, ops
    n, hw, c = 32, 128 * 128, 64
    torch.manual_seed(1)
    a = torch.randn(n, hw, c, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, hw, c, device="cuda").to(torch.bfloat16)
    first = None
    for _ in range(REPS):
        out = torch.zeros(n, c, device="cuda")
        ops.image_channel_dot(a, b, n, hw, c, 0.5, out, L.BF16)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(out, first)
    ref = 0.5 * (a.double() * b.double()).sum(1)
    assert float((first.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_msssim_loss_and_gradient_identical_on_every_launch():
    """MS-SSIM + L1 at the c2 loss shape (32 x 1 x 512 x 512): loss value and input gradient bit for bit on every evaluation."""
    from pssr2_amd.util import SSIMLoss
    g = torch.Generator().manual_seed(4)
    x = torch.rand(32, 1, 512, 512, generator=g).cuda()
    y = (x + 0.1 * torch.randn(32, 1, 512, 512, generator=g).cuda()).clamp(0, 1)
    loss_fn = SSIMLoss(mix=0.8)
    first = None
    for _ in range(REPS):
        xi = x.clone().requires_grad_(True)
        loss = loss_fn(xi, y)
        loss.backward()
        torch.cuda.synchronize()
        cur = (loss.detach().clone(), xi.grad.clone())
        if first is None:
            first = cur
        else:
            assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])
    assert torch.isfinite(first[0]) and float(first[1].abs().max()) > 0


def test_fused_adamw_identical_on_every_run():
    from pssr2_amd import ops
    n = 1 << 22
    g = torch.Generator().manual_seed(5)
    p0, gr = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda() * 1e-2
    first = None
    for _ in range(REPS):
        p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        for step in (1, 2, 3):
            ops.adamw_step(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step)
        torch.cuda.synchronize()
        if first is None:
            first = (p.clone(), m.clone(), v.clone())
        else:
            assert all(torch.equal(a, b) for a, b in zip((p, m, v), first))
