"""Oracle: SSIM / MS-SSIM + Gaussian-weighted L1 loss.  TEST INFRASTRUCTURE ONLY.

Restates
  * pssr/util.py:10-52 (SSIMLoss: ``mix*(1-ssim) + (1-mix)*mean(G (*) |x-y|)``), and
  * the third-party pytorch_msssim 1.0.0 algorithm it calls (pssr/util.py:5,30; pinned only
    as ``^1.0.0`` in pyproject.toml:33; source absent from /root/reference and this image):
    separable *valid* Gaussian filtering (H then W), K=(0.01,0.03), 5-level MS-SSIM with
    weights [0.0448, 0.2856, 0.3001, 0.2363, 0.1333], relu on cs / final ssim,
    avg_pool2d(k=2, padding=size%2) between levels, product of powers, mean over (N, C).
SSIM term: **parity unpinned** (no reference vector exists); see tests/test_oracle_loss.py for
the independent scipy cross-check.  The Gaussian-L1 term is pinned by tests/golden/loss_l1.npz.

Deviation (documented in DESIGN.md): for channels > 1 the reference's L1 window of shape
[1,1,k,k] with groups=C raises in torch; here the window is depthwise
[C,1,k,k], the evident intent.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def gauss_1d(size: int, sigma: float) -> torch.Tensor:
    coords = torch.arange(size, dtype=torch.float) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter_valid(x, g):
    c = x.shape[1]
    k = g.numel()
    out = x
    if x.shape[2] >= k:
        out = F.conv2d(out, g.view(1, 1, k, 1).repeat(c, 1, 1, 1), groups=c)
    if x.shape[3] >= k:
        out = F.conv2d(out, g.view(1, 1, 1, k).repeat(c, 1, 1, 1), groups=c)
    return out


def _ssim_maps(x, y, g, data_range, K=(0.01, 0.03)):
    c1 = (K[0] * data_range) ** 2
    c2 = (K[1] * data_range) ** 2
    g = g.to(x.dtype)
    mu1, mu2 = _filter_valid(x, g), _filter_valid(y, g)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _filter_valid(x * x, g) - mu1_sq
    s2 = _filter_valid(y * y, g) - mu2_sq
    s12 = _filter_valid(x * y, g) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = ((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)   # (N, C) each


def ssim(x, y, data_range=255.0, win_size=11, win_sigma=1.5, size_average=True):
    s, _ = _ssim_maps(x, y, gauss_1d(win_size, win_sigma), data_range)
    return s.mean() if size_average else s.mean(1)


def ms_ssim(x, y, data_range=255.0, win_size=11, win_sigma=1.5, size_average=True, weights=MS_WEIGHTS):
    assert min(x.shape[-2:]) > (win_size - 1) * 2 ** 4, "image too small for 5-level MS-SSIM"
    g = gauss_1d(win_size, win_sigma)
    w = torch.tensor(weights, dtype=x.dtype)
    vals = []
    for lvl in range(len(weights)):
        s, cs = _ssim_maps(x, y, g, data_range)
        if lvl < len(weights) - 1:
            vals.append(torch.relu(cs))
            pad = [d % 2 for d in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
    vals.append(torch.relu(s))
    stack = torch.stack(vals, dim=0)                       # (L, N, C)
    out = torch.prod(stack ** w.view(-1, 1, 1), dim=0)     # (N, C)
    return out.mean() if size_average else out.mean(1)


def gaussian_l1(x, y, win_size=11, win_sigma=1.5):
    """pssr/util.py:32-39,50 — mean over all pixels of the zero-padded G (*) |x-y| map."""
    c = x.shape[1]
    g = gauss_1d(win_size, win_sigma)
    win = torch.outer(g, g)[None, None].repeat(c, 1, 1, 1).to(x.dtype)
    return F.conv2d((x - y).abs(), win, groups=c, padding=(win_size - 1) // 2).mean()


def ssim_loss(x, y, mix=0.8, win_size=11, win_sigma=1.5, ms=True):
    """SSIMLoss.forward (pssr/util.py:45-52) with data_range=1 (pssr/util.py:30)."""
    s = ms_ssim(x, y, 1.0, win_size, win_sigma) if ms else ssim(x, y, 1.0, win_size, win_sigma)
    loss = 1 - s
    if mix < 1:
        loss = mix * loss + (1 - mix) * gaussian_l1(x, y, win_size, win_sigma)
    return loss
