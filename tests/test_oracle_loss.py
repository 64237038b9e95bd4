"""SSIM / MS-SSIM oracle: parity UNPINNED upstream (pytorch_msssim absent) — cross-checked here against
an independent float64 scipy.ndimage formulation and analytic cases."""
import numpy as np
import torch
from scipy.ndimage import correlate1d

from oracle import loss_ref as L


def _ssim_scipy(x, y, data_range, k=11, sigma=1.5):
    c = np.arange(k) - k // 2
    g = np.exp(-(c ** 2) / (2 * sigma ** 2)); g /= g.sum()
    r = k // 2

    def f(a):
        a = correlate1d(correlate1d(a.astype(np.float64), g, axis=-2, mode="constant"), g, axis=-1, mode="constant")
        return a[..., r:-r, r:-r]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mx, my = f(x), f(y)
    sxx, syy, sxy = f(x * x) - mx * mx, f(y * y) - my * my, f(x * y) - mx * my
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    return (((2 * mx * my + c1) / (mx * mx + my * my + c1)) * cs).mean((-2, -1)), cs.mean((-2, -1))


def _msssim_scipy(x, y, data_range):
    w = np.array(L.MS_WEIGHTS)
    vals = []
    for lvl in range(5):
        s, cs = _ssim_scipy(x, y, data_range)
        if lvl < 4:
            vals.append(np.maximum(cs, 0))
            assert x.shape[-1] % 2 == 0
            x = x.reshape(*x.shape[:-2], x.shape[-2] // 2, 2, x.shape[-1] // 2, 2).mean((-3, -1))
            y = y.reshape(*y.shape[:-2], y.shape[-2] // 2, 2, y.shape[-1] // 2, 2).mean((-3, -1))
    vals.append(np.maximum(s, 0))
    return np.prod(np.stack(vals) ** w.reshape(-1, 1, 1), 0).mean()


def test_ssim_vs_scipy():
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (2, 1, 64, 48)).astype(np.float32)
    y = np.clip(x + rng.normal(0, 0.1, x.shape), 0, 1).astype(np.float32)
    ref, _ = _ssim_scipy(x, y, 1.0)
    got = L.ssim(torch.tensor(x), torch.tensor(y), 1.0)
    assert abs(got.item() - ref.mean()) < 2e-6


def test_msssim_vs_scipy_and_identity():
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 1, (2, 1, 192, 192)).astype(np.float32)
    y = np.clip(x * 0.8 + rng.normal(0, 0.05, x.shape), 0, 1).astype(np.float32)
    got = L.ms_ssim(torch.tensor(x), torch.tensor(y), 1.0)
    assert abs(got.item() - _msssim_scipy(x, y, 1.0)) < 5e-6
    assert abs(L.ms_ssim(torch.tensor(x), torch.tensor(x), 1.0).item() - 1.0) < 1e-6
    assert abs(L.ssim_loss(torch.tensor(x), torch.tensor(x)).item()) < 1e-6


def test_loss_mix_and_grad():
    torch.manual_seed(0)
    x = torch.rand(1, 1, 176, 176, requires_grad=True)
    y = torch.rand(1, 1, 176, 176)
    full = L.ssim_loss(x, y, mix=0.8)
    parts = 0.8 * (1 - L.ms_ssim(x, y, 1.0)) + 0.2 * L.gaussian_l1(x, y)
    assert abs(full.item() - parts.item()) < 1e-7
    full.backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0
