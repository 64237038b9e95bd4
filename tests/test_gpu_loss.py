"""HIP SSIM / MS-SSIM + L1 loss and FusedAdamW vs the CPU oracle / torch.optim.AdamW."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [dict(), dict(mix=1.0), dict(mix=0.0), dict(ms=False), dict(ms=False, mix=0.3, win_size=7, win_sigma=1.0)])
@pytest.mark.parametrize("shape", [(2, 1, 192, 176), (1, 3, 177, 200)])
def test_ssim_loss_forward_backward(cfg, shape):
    from oracle import loss_ref as Lr
    from pssr2_amd.util import SSIMLoss
    g = torch.Generator().manual_seed(7)
    y = torch.rand(*shape, generator=g)
    x = (y * 0.7 + 0.3 * torch.rand(*shape, generator=g)).requires_grad_(True)
    ref = Lr.ssim_loss(x, y, mix=cfg.get("mix", 0.8), win_size=cfg.get("win_size", 11), win_sigma=cfg.get("win_sigma", 1.5),
                       ms=cfg.get("ms", True))
    ref.backward()
    xg = x.detach().cuda().requires_grad_(True)
    loss = SSIMLoss(channels=shape[1], **cfg)(xg, y.cuda())
    assert loss.shape == ()
    (loss * 1.5).backward()
    assert abs(loss.item() - ref.item()) < 2e-6 + 1e-5 * abs(ref.item())
    gref = x.grad.numpy() * 1.5
    np.testing.assert_allclose(xg.grad.cpu().numpy(), gref, rtol=2e-3, atol=2e-4 * np.abs(gref).max())


def test_l1_term_against_reference_fixture(golden):
    from pssr2_amd.util import SSIMLoss
    g = golden("loss_l1.npz")
    # mix=0 needs the SSIM value too; isolate L1 via two mixes: L(mix) = mix*s + (1-mix)*l1
    for i in (0, 1):
        x, y = torch.tensor(g[f"l1_{i}_x"]).cuda(), torch.tensor(g[f"l1_{i}_y"]).cuda()
        if min(x.shape[-2:]) < 11:
            continue
        a = SSIMLoss(mix=0.0, ms=False)(x, y).item()
        assert abs(a - float(g[f"l1_{i}_val"])) < 1e-6
        xg = x.clone().requires_grad_(True)
        SSIMLoss(mix=0.0, ms=False)(xg, y).backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), g[f"l1_{i}_grad"], rtol=1e-4, atol=1e-9)


def test_loss_size_checks():
    from pssr2_amd.util import SSIMLoss
    x = torch.rand(1, 1, 128, 128, device="cuda")
    with pytest.raises(AssertionError):
        SSIMLoss()(x, x)            # <= 160: MS-SSIM needs larger images, as pytorch_msssim asserts
    with pytest.raises(RuntimeError, match="MI355X"):
        SSIMLoss()(x.cpu(), x.cpu())


def test_fused_adamw_matches_torch():
    from pssr2_amd.optim import FusedAdamW
    torch.manual_seed(0)
    shapes = [(64, 16, 3, 3), (64,), (7, 5), (1,)]
    ref_p = [torch.randn(*s, device="cuda").requires_grad_(True) for s in shapes]
    my_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref = torch.optim.AdamW(ref_p, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.05)
    mine = FusedAdamW(my_p, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.05)
    for it in range(5):
        for a, b in zip(ref_p, my_p):
            gr = torch.randn_like(a)
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step(), mine.step()
    for a, b in zip(ref_p, my_p):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_forward_divided_equals_loss_of_quotients():
    """SSIMLoss.forward_divided(x, y, 255) is SSIMLoss()(x / 255, y / 255) (pssr/train.py:101) with the divisions done on load inside the
    training kernels: same loss and, for the network output, the same gradient bit for bit (IEEE division both ways)."""
    from pssr2_amd.util import SSIMLoss
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(3, 1, 192, 200, generator=g) * 255).cuda()
    y = (x + torch.randn(3, 1, 192, 200, generator=g).cuda() * 20).clamp(0, 255)
    lf = SSIMLoss(mix=0.8)
    xa = x.clone().requires_grad_(True)
    la = lf(xa / 255, y / 255)
    la.backward()
    xb = x.clone().requires_grad_(True)
    lb = lf.forward_divided(xb, y, 255)
    lb.backward()
    assert la.item() == lb.item()
    assert torch.equal(xa.grad, xb.grad)
    # windows the training kernels do not cover fall back to the quotient tensors
    lf7 = SSIMLoss(ms=False, win_size=7)
    xc = x.clone().requires_grad_(True)
    lc = lf7.forward_divided(xc, y, 255)
    lc.backward()
    xd = x.clone().requires_grad_(True)
    ld = lf7(xd / 255, y / 255)
    ld.backward()
    assert lc.item() == ld.item() and torch.equal(xc.grad, xd.grad)
