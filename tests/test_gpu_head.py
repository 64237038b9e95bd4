"""Dedicated Reconstruction.conv kernels (csrc/head_conv.hip, bf16 storage, blocked pixel order) vs torch-CPU F.conv2d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _blocked(x_nchw, blk, dt):
    """NCHW -> NHWC in the blocked pixel order of an r-times pixel-shuffle (r = 1 << blk)."""
    n, c, h, w = x_nchw.shape
    r = 1 << blk
    t = x_nchw.permute(0, 2, 3, 1).reshape(n, h // r, r, w // r, r, c).permute(0, 1, 3, 2, 4, 5)
    return t.reshape(n, h, w, c).contiguous().to(dt).cuda()


def _unblocked(t, blk):
    n, h, w, c = t.shape
    r = 1 << blk
    return t.float().cpu().reshape(n, h // r, w // r, r, r, c).permute(0, 1, 3, 2, 4, 5).reshape(n, h, w, c).permute(0, 3, 1, 2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 64, 1, 48, 80, 2), (1, 32, 3, 40, 24, 2), (3, 64, 2, 16, 16, 0), (1, 96, 1, 36, 20, 1)])
def test_head_conv(case, dt):
    from pssr2_amd import ops, _lib as L
    n, cin, cout, h, w, blk = case
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(cin + cout)
    act = F.relu(torch.randn(n, cin, h, w, generator=g)).to(dt).float().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5)
    wq = wt.to(dt).float().requires_grad_(True)
    b = torch.randn(cout, generator=g)
    ref = (F.conv2d(act, wq, b, padding=1)) * 128 + 128
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    assert ops.head_conv_supported(code, cin, cout)
    ad = _blocked(act.detach(), blk, dt)
    out = torch.empty(n, cout, h, w, device="cuda")
    ops.head_conv_fwd(ad, blk, wt.cuda().contiguous(), b.cuda(), out, n, h, w, cin, cout, 128.0, 128.0, code)
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), rtol=2e-3, atol=2e-2 * 128 * ref.detach().sub(128).abs().max().item() / 128)
    # dgrad with the ReLU mask of the activation
    da = torch.full_like(ad, 7.0)
    ops.head_conv_dgrad(dout.cuda(), 128.0, wt.cuda().contiguous(), ad, da, blk, n, h, w, cin, cout, code)
    want = act.grad * (act.detach() > 0)
    got = _unblocked(da, blk)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=3e-2, atol=2e-2 * want.abs().max().item())
    # wgrad, accumulated into an existing gradient
    base = torch.randn(cout, cin, 3, 3, generator=g)
    dw = base.clone().cuda()
    ops.head_conv_wgrad(dout.cuda(), 128.0, ad, blk, dw, n, h, w, cin, cout, code)
    np.testing.assert_allclose((dw.cpu() - base).numpy(), wq.grad.numpy(), rtol=1e-3, atol=1e-3 * wq.grad.abs().max().item())
    # dgrad + wgrad + bias sums in one pass (the path the engine takes): same dact, dW from bf16/fp16 MFMA operands
    if blk <= 2:
        da2 = torch.full_like(ad, 7.0)
        dw2 = base.clone().cuda()
        r = 1 << blk
        bsum = torch.zeros(r * r * cin, device="cuda") if cin in (32, 64, 128) else None
        ops.head_conv_bwd(dout.cuda(), 128.0, wt.cuda().contiguous(), ad, da2, blk, dw2, bsum, n, h, w, cin, cout, code)
        assert torch.equal(da2, da)
        np.testing.assert_allclose((dw2.cpu() - base).numpy(), wq.grad.numpy(), rtol=2e-2, atol=1e-2 * wq.grad.abs().max().item())
        if bsum is not None:
            want_b = da.float().cpu().reshape(n, h // r, w // r, r * r * cin).sum(dim=(0, 1, 2))
            np.testing.assert_allclose(bsum.cpu().numpy(), want_b.numpy(), rtol=1e-3, atol=1e-3 * want_b.abs().max().item())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 32, 32), (1, 36, 52), (3, 16, 16)])
def test_inference_head_inside_pre_epilogue(monkeypatch, dt, shape):
    """Round 4: in eval mode Reconstruction.pre's epilogue forms the nine tap products of Reconstruction.conv out of its accumulators
    (PSSR_EPI_HEADQ) and pssr_head_q_gather sums the shifted taps -- the pixel-shuffled 64-channel tensor is never stored.  Against the
    stored-activation path (PSSR_HEAD_FUSE=0) on the same weights: the same 16-bit products, f32 sums in another order."""
    import pssr2_amd.engine as E
    from pssr2_amd.models import ResUNet
    n, h, w = shape
    torch.manual_seed(h + w)
    model = ResUNet(hidden=[64, 128], depth=1).cuda().eval()
    model.infer_dtype = dt
    with torch.no_grad():
        model.reconstruction.pre.bias.normal_(0, 0.2)
        model.reconstruction.conv.weight.mul_(3.0)
        model.reconstruction.conv.bias.fill_(0.1)
    x = (torch.rand(n, 1, h, w) * 255).cuda()
    outs = {}
    for fuse in (True, False):
        monkeypatch.setattr(E, "_HEAD_FUSE", fuse)
        with torch.no_grad():
            outs[fuse] = model(x).float().clone()
        plan = list(model._engine.plans.values())[-1]
        assert (getattr(plan, "head_q", None) is not None) == fuse or not fuse
    a, b = outs[True], outs[False]
    assert a.shape == (n, 1, 4 * h, 4 * w) and torch.isfinite(a).all()
    err = float((a - b).abs().max())
    assert err <= 2e-3 * max(1.0, float(b.abs().max()) / 128), err
    ua, ub = a.clamp(0, 255).to(torch.uint8), b.clamp(0, 255).to(torch.uint8)
    assert float((ua != ub).float().mean()) <= 1e-3 and int((ua.int() - ub.int()).abs().max()) <= 1


def test_training_head_keeps_the_activation_and_the_tap_planes(monkeypatch):
    """FLAG_HEADQ (training): `pre` stores its activation for the backward pass exactly as before (bit for bit) and emits the tap planes in
    the same launch; the network output equals the stored-activation head's up to f32 summation order, and so do the gradients."""
    import pssr2_amd.engine as E
    from pssr2_amd.models import ResUNet
    torch.manual_seed(5)
    x = (torch.rand(2, 1, 32, 32) * 255).cuda()
    tgt = (torch.rand(2, 1, 128, 128) * 255).cuda()
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(E, "_HEAD_FUSE", fuse)
        torch.manual_seed(7)
        model = ResUNet(hidden=[64, 128], depth=1).cuda().train()
        model.compute_dtype = torch.bfloat16
        y = model(x)
        torch.nn.functional.mse_loss(y / 255, tgt / 255).backward()
        torch.cuda.synchronize()
        plan = list(model._engine.plans.values())[-1]
        res[fuse] = (y.detach().float().clone(), plan.pre.clone(), {n: p.grad.clone() for n, p in model.named_parameters()})
    (ya, pa, ga), (yb, pb, gb) = res[True], res[False]
    assert torch.equal(pa, pb)
    assert float((ya - yb).abs().max()) <= 2e-3 * max(1.0, float(yb.abs().max()) / 128)
    for n in ga:
        scale = float(gb[n].abs().max()) + 1e-12
        assert float((ga[n] - gb[n]).abs().max()) <= 2e-2 * scale, n


@pytest.mark.parametrize("shape", [(2, 16, 20), (3, 17, 18), (1, 16, 16), (2, 5, 128)])
def test_head_q_gather_sums_the_shifted_tap_planes(shape):
    """pssr_head_q_gather alone, both kernels (widths that are / are not multiples of 4): out[P] = bias + sum over the nine taps of the
    plane value at P + (ky - 1, kx - 1), zero outside the image, against the same sum in float64."""
    from pssr2_amd import ops
    n, h, w = shape
    torch.manual_seed(n * 100 + w)
    q = torch.randn(9, 16, n, h, w, device="cuda")
    bias = torch.tensor([0.3], device="cuda")
    out = torch.full((n, 1, 4 * h, 4 * w), float("nan"), device="cuda")
    ops.head_q_gather(q, bias, out, n, h, w, 2.0, -1.0)
    hr = q.double().reshape(9, 4, 4, n, h, w).permute(0, 3, 4, 1, 5, 2).reshape(9, n, 4 * h, 4 * w)       # [tap][n][4y+i][4x+j]
    pad = torch.nn.functional.pad(hr, (1, 1, 1, 1))
    want = torch.full((n, 4 * h, 4 * w), 0.3, dtype=torch.float64, device="cuda")
    for ky in range(3):
        for kx in range(3):
            want += pad[ky * 3 + kx, :, ky:ky + 4 * h, kx:kx + 4 * w]
    want = want * 2.0 - 1.0
    assert float((out[:, 0].double() - want).abs().max()) <= 1e-5
