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
        scaled = resize(hr_hat_norm, hr_norm.shape) if hr_hat_norm.shape != hr_norm.shape else hr_hat_norm      # pssr/util.py:179
        amp = np.cov(scaled.flatten(), hr_norm.flatten())[0, 1] / np.var(hr_hat_norm.flatten())
        hr_hat_norm = amp * hr_hat_norm
        hr_norm, hr_hat_norm = (hr_norm - hr_norm.min()) * base_max, (hr_hat_norm - hr_norm.min()) * base_max
        hr_norm, hr_hat_norm = hr_norm / (hr_norm.mean() / base_mean), hr_hat_norm / (hr_hat_norm.mean() / base_mean)
        hr_norms.append(hr_norm)
        hr_hat_norms.append(hr_hat_norm)
    hr, hr_hat = np.asarray(hr_norms).clip(0, 255), np.asarray(hr_hat_norms).clip(0, 255)
    return hr.reshape(hr_shape).astype(np.uint8), hr_hat.reshape(hr_hat_shape).astype(np.uint8)


def resize(image, output_shape):
    """skimage.transform.resize(image, output_shape) with its defaults (order=1, mode="reflect", cval=0, clip=True,
    preserve_range=False, anti_aliasing=True) for a 2-D float image, as scikit-image >= 0.19 computes it (scikit-image is absent
    from this image and from the reference tree: PARITY UNPINNED, restated from the published source of
    skimage/transform/_warps.py): a Gaussian pre-filter with sigma = max(0, (input / output - 1) / 2) per axis when any axis
    shrinks (scipy.ndimage.gaussian_filter, mode "mirror" = skimage's "reflect", cval 0), then
    scipy.ndimage.zoom(order=1, mode="mirror", grid_mode=True) -- output pixel centres mapped onto the input grid by
    (i + 0.5) * input / output - 0.5 -- and a clip to the input's value range.  scipy IS the engine scikit-image calls here."""
    from scipy import ndimage as ndi
    image = np.asarray(image)
    if image.dtype not in (np.float32, np.float64):
        image = image.astype(np.float64)
    in_shape, out_shape = np.array(image.shape, float), np.array(output_shape, float)
    factors = in_shape / out_shape
    filtered = image
    if np.any(out_shape < in_shape):                      # anti_aliasing=None -> True for a float image that shrinks along an axis
        sigma = np.maximum(0, (factors - 1) / 2)
        filtered = ndi.gaussian_filter(image, sigma, cval=0, mode="mirror")
    out = ndi.zoom(filtered, 1 / factors, order=1, mode="mirror", cval=0, grid_mode=True)
    assert out.shape == tuple(output_shape)
    return np.clip(out, image.min(), image.max())         # _clip_warp_output (clip=True; mode "reflect" does not extend the range)


def resize_restated(image, output_shape):
    """The same resize written out element by element (what csrc/metrics.hip implements), checked against ``resize`` -- i.e. against
    scipy -- by tests/test_oracle_metrics.py: separable Gaussian (radius int(4 sigma + 0.5), weights exp(-t^2 / (2 sigma^2)) normalised,
    mirror boundary, float64 accumulation, rounded to the image type after each axis, axis 0 first), then linear interpolation
    between floor(c) and floor(c) + 1 at c = (i + 0.5) * in / out - 0.5 with mirrored indices, in float64, rounded to the image type."""
    image = np.asarray(image)
    dt = image.dtype if image.dtype in (np.float32, np.float64) else np.float64
    img = image.astype(dt)

    def mirror(i, n):
        if n == 1:
            return np.zeros_like(i)
        period = 2 * (n - 1)
        i = np.abs(i) % period
        return np.where(i >= n, period - i, i)

    def gauss_axis(a, sigma, axis):
        if sigma <= 0:
            return a
        r = int(4.0 * sigma + 0.5)
        t = np.arange(-r, r + 1)
        wts = np.exp(-0.5 / (sigma * sigma) * t.astype(np.float64) ** 2)
        wts /= wts.sum()
        n = a.shape[axis]
        idx = mirror(np.arange(n)[:, None] + t[None, :], n)          # [n, 2r+1]
        src = np.moveaxis(a, axis, 0).astype(np.float64)
        out = np.einsum("ik,ik...->i...", wts[None, :].repeat(n, 0), src[idx])
        return np.moveaxis(out, 0, axis).astype(dt)

    in_shape, out_shape = img.shape, tuple(output_shape)
    if any(o < i for o, i in zip(out_shape, in_shape)):
        for ax in range(2):
            img = gauss_axis(img, max(0.0, (in_shape[ax] / out_shape[ax] - 1) / 2), ax)

    def axis_map(n_in, n_out):
        c = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        f = np.floor(c)
        return mirror(f.astype(np.int64), n_in), mirror(f.astype(np.int64) + 1, n_in), c - f
    y0, y1, ty = axis_map(in_shape[0], out_shape[0])
    x0, x1, tx = axis_map(in_shape[1], out_shape[1])
    v = img.astype(np.float64)
    out = ((1 - ty)[:, None] * (1 - tx)[None, :] * v[y0][:, x0] + (1 - ty)[:, None] * tx[None, :] * v[y0][:, x1]
           + ty[:, None] * (1 - tx)[None, :] * v[y1][:, x0] + ty[:, None] * tx[None, :] * v[y1][:, x1])
    return np.clip(out.astype(dt), image.min(), image.max())


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
