"""Crappifiers: the reference's degradation transforms (pssr/crappifiers.py) with two call paths.

* ``crappify(np.ndarray) -> np.ndarray``: the reference contract (numpy in / numpy out, global legacy
  ``np.random`` stream), so user subclasses and DataLoader workers behave as they do upstream;
* ``crappify_device(lr_u8_tensor, seed) -> float32 tensor``: the MI355X path used by the device-side
  pair generator (pssr2_amd.data.DevicePairGenerator): counter-based Philox noise keyed by
  (seed, tile, pixel), fused with the round-half-even + clip of pssr/data.py:487.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np


class Crappifier(ABC):
    r"""Base class for custom crappifiers: override :meth:`crappify` (pssr/crappifiers.py:6-24)."""

    @abstractmethod
    def crappify(self, image: np.ndarray):
        raise NotImplementedError('"crappify" method not implemented.')

    def __call__(self, image: np.ndarray):
        return self.crappify(image)

    # device-side description consumed by csrc/crappify.hip: (kind, intensity, gain, spread)
    def device_spec(self):
        raise NotImplementedError(f"{type(self).__name__} has no MI355X device path; it runs on the host through crappify()")


def _draw_intensity(mean, spread):
    return max(np.random.normal(mean, spread), 0) if spread > 0 else mean


class MultiCrappifier(Crappifier):
    def __init__(self, *args: Crappifier, clip: bool = True):
        r"""Applies crappifiers in order, clipping to [0, 255] after each when ``clip`` (pssr/crappifiers.py:26-43)."""
        self.crappifiers = args
        self.clip = clip

    def crappify(self, image: np.ndarray):
        for stage in self.crappifiers:
            image = stage.crappify(image)
            if self.clip:
                image = np.clip(image, 0, 255)
        return image

    def device_spec(self):
        return [("clip" if self.clip else "noclip",)] + [c.device_spec() for c in self.crappifiers]


class AdditiveGaussian(Crappifier):
    def __init__(self, intensity: float = 13, gain: float = 0, spread: float = 0):
        r"""Adds N(gain, intensity) noise; ``spread`` randomises the intensity per call (pssr/crappifiers.py:45-64)."""
        self.intensity, self.gain, self.spread = intensity, gain, spread

    def crappify(self, image: np.ndarray):
        sigma = _draw_intensity(self.intensity, self.spread)          # scalar draw first, then the field
        return image.astype(np.float32) + np.random.normal(self.gain, sigma, image.shape)

    def device_spec(self):
        return ("gaussian", float(self.intensity), float(self.gain), float(self.spread))


class Poisson(Crappifier):
    def __init__(self, intensity: float = 1, gain: float = 0, spread: float = 0):
        r"""Shot noise: x*(1-i) + Poisson(x)*i + gain (pssr/crappifiers.py:66-86)."""
        self.intensity, self.gain, self.spread = intensity, gain, spread

    def crappify(self, image: np.ndarray):
        draw = np.random.poisson(np.clip(image, 0, np.inf))           # field first, then the scalar
        i = _draw_intensity(self.intensity, self.spread)
        return image.astype(np.float32) * (1 - i) + draw * i + self.gain

    def device_spec(self):
        return ("poisson", float(self.intensity), float(self.gain), float(self.spread))


class SaltPepper(Crappifier):
    def __init__(self, intensity: float = 0.5, gain: float = 0, spread: float = 0):
        r"""Replaces ``intensity`` percent of the pixels by 0 or 255 (pssr/crappifiers.py:88-105).

        Restates skimage.util.random_noise(mode="s&p") — scikit-image is not a dependency here:
        Bernoulli(amount) flip mask, Bernoulli(0.5) salt mask, both from ``np.random.default_rng()``.
        """
        self.intensity, self.gain, self.spread = intensity / 100, gain, spread

    def crappify(self, image: np.ndarray):
        amount = _draw_intensity(self.intensity, self.spread)
        out = np.clip(image.astype(np.float32) + self.gain, 0, 255) / 255
        rng = np.random.default_rng()
        flipped = rng.random(out.shape) <= amount
        salted = rng.random(out.shape) <= 0.5
        out = out.astype(np.float64)
        out[flipped & salted] = 1
        out[flipped & ~salted] = 0
        return np.clip(out, 0, 1) * 255

    def device_spec(self):
        return ("saltpepper", float(self.intensity), float(self.gain), float(self.spread))


class Blur(Crappifier):
    def __init__(self, intensity: float = 2, gain: float = 0, spread: float = 0):
        r"""Gaussian blur of sigma ``intensity`` per frame, edge-replicated, radius int(4*sigma+.5)
        (pssr/crappifiers.py:107-124 -> skimage.filters.gaussian -> scipy.ndimage.gaussian_filter)."""
        self.intensity, self.gain, self.spread = intensity, gain, spread

    def crappify(self, image: np.ndarray):
        sigma = _draw_intensity(self.intensity, self.spread)
        img = image.astype(np.float32)
        if sigma <= 0:
            return img + self.gain
        r = int(4.0 * float(sigma) + 0.5)
        x = np.arange(-r, r + 1, dtype=np.float64)
        w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
        w /= w.sum()
        out = img
        for axis in (1, 2):
            n = out.shape[axis]
            padded = np.take(out, np.clip(np.arange(-r, n + r), 0, n - 1), axis=axis).astype(np.float64)
            acc = np.zeros(out.shape, dtype=np.float64)
            for k in range(2 * r + 1):
                sl = [slice(None)] * 3
                sl[axis] = slice(k, k + n)
                acc += w[k] * padded[tuple(sl)]
            out = acc.astype(np.float32)
        return out + self.gain

    def device_spec(self):
        return ("blur", float(self.intensity), float(self.gain), float(self.spread))
