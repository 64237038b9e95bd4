"""CPU restatement (TEST INFRASTRUCTURE, never imported by pssr2_amd/) of the reference's prediction normalisation and image
metrics: pssr/util.py:139-218 and the skimage metrics pssr/predict.py:199-202 calls.

* normalize_preds / _normalize_minmax follow pssr/util.py:139-205 operation by operation (same numpy calls and dtypes) and are
  PINNED bit-exactly by tests/golden/metrics.npz (outputs of the reference itself, oracle/gen_golden.py:gen_metrics).
* psnr / ssim restate skimage.metrics.peak_signal_noise_ratio / structural_similarity (scikit-image is absent from this image
  and from the reference tree): PARITY UNPINNED, written from the published definitions (Wang et al. 2004; uniform 7x7 window,
  K1 = 0.01, K2 = 0.03, sample covariance, mean over the interior)."""
import numpy as np


def _normalize_minmax(x, pmin=0.1, pmax=99.9, eps=1e-20, dtype=np.float32):      # pssr/util.py:193-205 (csbdeep)
    x_min = np.percentile(x, pmin, keepdims=True)
    x_max = np.percentile(x, pmax, keepdims=True)
    x = x.astype(dtype, copy=False)
    x_min = dtype(x_min) if np.isscalar(x_min) else x_min.astype(dtype, copy=False)
    x_max = dtype(x_max) if np.isscalar(x_max) else x_max.astype(dtype, copy=False)
    eps = dtype(eps)
    return (x - x_min) / (x_max - x_min + eps)


def normalize_preds(hr, hr_hat, pmin=0.1, pmax=99.9):                              # pssr/util.py:139-191
    hr, hr_hat = np.asarray(hr), np.asarray(hr_hat)
    if len(hr.shape) != len(hr_hat.shape):
        raise ValueError("hr and hr_hat must have the same number of dimensions")
    hr_shape, hr_hat_shape = hr.shape, hr_hat.shape
    if len(hr.shape) < 3:
        hr, hr_hat = hr[np.newaxis, ...], hr_hat[np.newaxis, ...]
    hr, hr_hat = hr.reshape(-1, *hr.shape[-2:]), hr_hat.reshape(-1, *hr_hat.shape[-2:])
    if len(hr) != len(hr_hat):
        raise ValueError("hr and hr_hat must have the same number of images")
    hr_norms, hr_hat_norms = [], []
    for idx in range(len(hr)):
        hr_norm = hr[idx].astype(np.float32)
        hr_hat_norm = hr_hat[idx].astype(np.float32)
        base_max = np.percentile(hr_norm, pmax)
        base_mean = np.mean(hr_norm)
        hr_norm = _normalize_minmax(hr_norm, pmin, pmax)
        hr_hat_norm = hr_hat_norm - np.mean(hr_hat_norm)
        hr_norm = hr_norm - np.mean(hr_norm)
        if hr_hat_norm.shape != hr_norm.shape:
            raise NotImplementedError("mismatched sizes go through skimage.transform.resize in the reference (pssr/util.py:176)")
        amp = np.cov(hr_hat_norm.flatten(), hr_norm.flatten())[0, 1] / np.var(hr_hat_norm.flatten())
        hr_hat_norm = amp * hr_hat_norm
        hr_norm, hr_hat_norm = (hr_norm - hr_norm.min()) * base_max, (hr_hat_norm - hr_norm.min()) * base_max
        hr_norm, hr_hat_norm = hr_norm / (hr_norm.mean() / base_mean), hr_hat_norm / (hr_hat_norm.mean() / base_mean)
        hr_norms.append(hr_norm)
        hr_hat_norms.append(hr_hat_norm)
    hr, hr_hat = np.asarray(hr_norms).clip(0, 255), np.asarray(hr_hat_norms).clip(0, 255)
    return hr.reshape(hr_shape).astype(np.uint8), hr_hat.reshape(hr_hat_shape).astype(np.uint8)


def psnr(a, b, data_range=255):
    err = np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)
    return 10 * np.log10(data_range ** 2 / err)


def ssim(a, b, data_range=255, win=7, k1=0.01, k2=0.03):
    from scipy.ndimage import uniform_filter
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    npx = win * win
    cov_norm = npx / (npx - 1)
    ux, uy = uniform_filter(a, win), uniform_filter(b, win)
    uxx, uyy, uxy = uniform_filter(a * a, win), uniform_filter(b * b, win), uniform_filter(a * b, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return s[pad:-pad, pad:-pad].mean()
