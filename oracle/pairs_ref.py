"""Oracle: LR/HR pair generation, crappifier arithmetic and prediction post-ops (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Integer paths are bit-exact restatements.

Follows
  * pssr/data.py:471-495   _gen_pair      (crop, reflect pad, rot90/flip, bilinear /scale, crappify, round/clip)
  * pssr/data.py:483       Pillow ``Image.resize(BILINEAR)`` on uint8 "L" images — third-party
                           (Pillow >=9.1, pyproject.toml:27): restated from Pillow's published
                           two-pass fixed-point resampler (precision 22 bits, triangle filter,
                           support scaled by the reduction ratio); pinned by tests/golden/bilinear.npz
  * pssr/data.py:536-551   _square_crop / _pad_image
  * pssr/data.py:662-668   _slice_center
  * pssr/crappifiers.py:45-124   AdditiveGaussian / Poisson / Blur / MultiCrappifier arithmetic
  * pssr/predict.py:245-246      _pred_array (clip then uint8 truncation)
  * pssr/util.py:116-137         _patch_images (overlap-average stitching)
  * pssr/data.py:629-638,682-687 _sliding_window / _n_tiles
  * pssr/data.py:708-735         _get_val_idx / _invert_idx
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2   # Pillow: 8-bit pixels, 22-bit fixed-point coefficients


# ----------------------------------------------------------------------------- bilinear (Pillow)
def pil_bilinear_coeffs(in_size: int, out_size: int):
    """Per-output (xmin, integer taps) of Pillow's BILINEAR reduction along one axis."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    bounds, taps = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size)
        x = np.arange(xmin, xmax, dtype=np.float64)
        w = np.maximum(0.0, 1.0 - np.abs((x + 0.5 - center) / filterscale))
        w = w / w.sum()
        k = np.where(w < 0, (w * (1 << PRECISION_BITS) - 0.5).astype(np.int64),
                     (w * (1 << PRECISION_BITS) + 0.5).astype(np.int64))
        bounds.append(xmin)
        taps.append(k)
    return bounds, taps


def _resample_axis_u8(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    img = np.moveaxis(img, axis, -1)
    bounds, taps = pil_bilinear_coeffs(img.shape[-1], out_size)
    out = np.empty(img.shape[:-1] + (out_size,), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx, (x0, k) in enumerate(zip(bounds, taps)):
        acc = (1 << (PRECISION_BITS - 1)) + (src[..., x0:x0 + len(k)] * k).sum(-1)
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def pil_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [..., H, W] -> uint8 [..., out_h, out_w]; horizontal pass, then vertical pass."""
    assert img.dtype == np.uint8
    tmp = _resample_axis_u8(img, out_w, -1) if img.shape[-1] != out_w else img
    return _resample_axis_u8(tmp, out_h, -2) if img.shape[-2] != out_h else tmp


# ----------------------------------------------------------------------------- geometry
def square_crop(image, max_res):
    h, w = image.shape[-2:]
    if [h, w] == [max_res] * 2:
        return image
    size = min(h, w, max_res)
    sx, sy = (h - size) // 2, (w - size) // 2
    return image[:, sx:sx + size, sy:sy + size]


def pad_image(image, res):
    if image.shape[-1] < res:
        p = res - image.shape[-1]
        return np.stack([np.pad(ch, [[0, p], [0, p]], mode="reflect") for ch in image])
    return image


def slice_center(image, n_frames):
    center, half = image.shape[-3] // 2, n_frames // 2
    if n_frames % 2 == 0:
        return image[..., center - half:center + half, :, :]
    return image[..., center - half:center + half + 1, :, :]


def augment(hr, rotation):
    """rotation = False | [rot90: bool, flip axis: 1 | 2 | (1, 2)] (pssr/data.py:108,478-480)."""
    if rotation:
        hr = np.rot90(hr, axes=(1, 2)) if rotation[0] else hr
        hr = np.flip(hr, axis=rotation[1])
    return hr


def round_clip(lr):
    """pssr/data.py:487 — np.round is round-half-to-even."""
    return np.clip(np.round(lr), 0, 255)


def gen_pair(hr_u8, hr_res, lr_scale, rotation, crappify, n_frames=None):
    """_gen_pair with the crappifier given as a callable on the float32 LR stack (or None)."""
    hr = pad_image(square_crop(hr_u8, hr_res), hr_res)
    hr = augment(hr, rotation)
    lr_res = hr_res // lr_scale
    lr = pil_bilinear_u8(np.ascontiguousarray(hr), lr_res, lr_res).astype(np.float32)
    if crappify is not None:
        lr = round_clip(crappify(lr))
    if n_frames is not None and n_frames[0] != n_frames[1]:
        if not n_frames[1] > hr.shape[-3]:
            hr = slice_center(hr, n_frames[1])
        if not n_frames[0] > lr.shape[-3]:
            lr = slice_center(lr, n_frames[0])
    return hr.astype(np.float32), lr.astype(np.float32)


# ----------------------------------------------------------------------------- crappifier arithmetic
def additive_gaussian(image, noise):
    """AdditiveGaussian.crappify with the N(gain, sigma) field supplied (float64)."""
    return image.astype(np.float32) + noise


def poisson_mix(image, draw, intensity=1.0, gain=0.0):
    """Poisson.crappify with the Poisson draw supplied: x*(1-i) + y*i + gain."""
    return image.astype(np.float32) * (1 - intensity) + draw * intensity + gain


def gaussian_blur_nearest(image, sigma, truncate=4.0):
    """Blur.crappify arithmetic: per-channel separable Gaussian, edge-replicate, radius int(4s+.5).

    skimage.filters.gaussian(float32 image, sigma, channel_axis=0) ->
    scipy.ndimage.gaussian_filter(sigma=(0, s, s), mode="nearest", truncate=4.0).
    """
    img = image.astype(np.float32)
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    w = w / w.sum()
    out = img
    for axis in (1, 2):   # rows (H) first, then columns (W): scipy filters axes in order
        n = out.shape[axis]
        idx = np.clip(np.arange(-r, n + r), 0, n - 1)
        padded = np.take(out, idx, axis=axis).astype(np.float64)
        acc = np.zeros(out.shape, dtype=np.float64)
        for k in range(2 * r + 1):
            sl = [slice(None)] * 3
            sl[axis] = slice(k, k + n)
            acc += w[k] * padded[tuple(sl)]
        out = acc.astype(np.float32)
    return out


# ----------------------------------------------------------------------------- prediction post-ops
def pred_array(x, n_frames=1):
    return slice_center(np.clip(x, 0, 255).astype(np.uint8), n_frames)


def n_tiles(shape_hw, size, stride):
    x, y = shape_hw
    return max(0, (x - size) // stride + 1), max(0, (y - size) // stride + 1)


def sliding_tile(image, size, stride, tile_idx):
    _, ty = n_tiles(image.shape[-2:], size, stride)
    sx, sy = tile_idx // ty * stride, tile_idx % ty * stride
    return image[..., sx:sx + size, sy:sy + size]


def patch_images(batched, n_cols, n_rows, overlap, margin):
    size = batched.shape[-1]
    step = size - overlap
    H, W = n_rows * step + overlap, n_cols * step + overlap
    collage, count = np.zeros((H, W)), np.zeros((H, W))
    for idx in range(n_rows * n_cols):
        row, col = idx // n_cols, idx % n_cols
        r0, c0 = row * step, col * step
        m = [margin if row != 0 else 0, margin if row != n_rows - 1 else 0,
             margin if col != 0 else 0, margin if col != n_cols - 1 else 0]
        collage[r0 + m[0]:r0 + size - m[1], c0 + m[2]:c0 + size - m[3]] += \
            batched[idx, m[0]:batched.shape[1] - m[1], m[2]:batched.shape[2] - m[3]]
        count[r0 + m[0]:r0 + size - m[1], c0 + m[2]:c0 + size - m[3]] += 1
    count[count == 0] = 1
    return collage / count


def get_val_idx(slices, split, seed, tiles=None):
    if tiles is not None:
        slices = [s for s, t in zip(slices, tiles) for _ in range(t)]
    order = list(range(len(slices)))
    if seed is not None and split < 1:
        np.random.seed(seed)
        np.random.shuffle(order)
    chosen = set(order[-max(1, int(split * len(slices))):])
    val, pos = [], 0
    for i, s in enumerate(slices):
        if i in chosen:
            val.extend(range(pos, pos + s))
        pos += s
    return val


def invert_idx(idx, n):
    r = np.arange(n)
    return r[np.logical_not(np.isin(r, idx))]
