"""Datasets and pair generation with the reference's protocol (pssr/data.py).

Two ways to produce (HR, LR) training pairs from uint8 HR tiles:

* host path — ``_gen_pair`` exactly as the reference orders it (crop, reflect-pad, rot90/flip, Pillow
  BILINEAR reduction, crappifier on numpy, round-half-even + clip): used by ``ImageDataset`` /
  ``ArrayDataset.__getitem__`` so that DataLoader workers and user crappifier subclasses keep working;
* device path — ``DevicePairGenerator``: whole batches of uint8 HR tiles resident in HBM go through
  the HIP kernels (bit-exact Pillow reduction, Philox noise, fused round/clip), removing the
  ~1 ms/tile host stage that would otherwise cap multi-GPU training.

File decoding (tif/czi) is out of scope: ``ImageDataset`` reads what Pillow reads.
"""
from __future__ import annotations

import glob
import random
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from .crappifiers import Crappifier, Poisson
from .util import _force_list


# --------------------------------------------------------------------------------------- geometry
def _square_crop(image, max_res):
    h, w = image.shape[-2:]
    if [h, w] == [max_res] * 2:
        return image
    size = min(h, w, max_res)
    sx, sy = (h - size) // 2, (w - size) // 2
    return image[:, sx:sx + size, sy:sy + size]


def _pad_image(image, res):
    if image.shape[-1] < res:
        p = res - image.shape[-1]
        return np.stack([np.pad(ch, [[0, p], [0, p]], mode="reflect") for ch in image])
    return image


def _slice_center(image, n_frames):
    center, half = image.shape[-3] // 2, n_frames // 2
    if n_frames % 2 == 0:
        return image[..., center - half:center + half, :, :]
    return image[..., center - half:center + half + 1, :, :]


def _tensor_ready(image, transforms, compact=False):
    """float32 tensor of an image (pssr/data.py:497-505).  ``compact``: uint8 instead -- what the drivers of this package ask their own
    datasets for while they feed a captured graph from a DataLoader (every value here is an integer in [0, 255]: uint8 pixels, or the
    rounded and clipped crappifier output); the conversion to float32 then happens on the device, and the worker -> pin-memory thread ->
    PCIe path moves a quarter of the bytes."""
    if compact and transforms is None:
        u8 = np.ascontiguousarray(image).astype(np.uint8)
        if image.dtype == np.uint8 or np.array_equal(u8, image):        # (a crappifier that returned NaN / a value outside [0, 255]: float32 as always)
            return torch.from_numpy(u8)
    t = torch.tensor(np.ascontiguousarray(image).astype(np.float32), dtype=torch.float)
    if transforms is not None:
        for tr in transforms:
            t = tr(t)
    return t


def _resize_bilinear_u8(hr, lr_res):
    """Per-frame ``PIL.Image.resize(BILINEAR)`` (pssr/data.py:483)."""
    from PIL import Image
    return np.stack([np.asarray(Image.fromarray(ch).resize([lr_res] * 2, Image.Resampling.BILINEAR)) for ch in hr])


def _gen_pair(hr, hr_res, lr_scale, rotation, crappifier, transforms, n_frames, compact=False):
    """Training pair from one uint8 HR stack [C, H, W] (pssr/data.py:471-495)."""
    hr = _pad_image(_square_crop(hr, hr_res), hr_res)
    if rotation:
        hr = np.rot90(hr, axes=(1, 2)) if rotation[0] else hr
        hr = np.flip(hr, axis=rotation[1])
    lr = _resize_bilinear_u8(np.ascontiguousarray(hr), hr_res // lr_scale).astype(np.float32)
    if crappifier is not None:
        lr = crappifier.crappify(lr) if issubclass(type(crappifier), Crappifier) else crappifier(lr)
        lr = np.clip(lr.round(), 0, 255)
    if n_frames is not None and n_frames[0] != n_frames[1]:
        if not n_frames[1] > hr.shape[-3]:
            hr = _slice_center(hr, n_frames[1])
        if not n_frames[0] > lr.shape[-3]:
            lr = _slice_center(lr, n_frames[0])
    return _tensor_ready(hr, transforms, compact), _tensor_ready(lr, transforms, compact)


def _ready_lr(lr, lr_res, transforms, compact=False):
    return _tensor_ready(_pad_image(_square_crop(lr, lr_res), lr_res), transforms, compact)


def _n_tiles(image, size, stride):
    x, y = image.shape[-2:]
    return max(0, (x - size) // stride + 1), max(0, (y - size) // stride + 1)


def _sliding_tile(image, size, stride, tile_idx):
    _, ty = _n_tiles(image, size, stride)
    sx, sy = tile_idx // ty * stride, tile_idx % ty * stride
    return image[..., sx:sx + size, sy:sy + size]


def _get_n_frames(n_frames):
    if n_frames in [None, -1, [-1]]:
        return None
    n_frames = _force_list(n_frames)
    return n_frames * 2 if len(n_frames) == 1 else n_frames


def _get_val_idx(slices, split, seed, tiles=None):
    """Validation frame indices (pssr/data.py:708-730): numpy legacy shuffle under ``seed``."""
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


def _invert_idx(idx, idx_len):
    r = np.arange(idx_len)
    return r[np.logical_not(np.isin(r, idx))]


class _RandomIterIdx:
    """Sampler of pssr/data.py:737-752; ``rank``/``world`` shard the epoch for data-parallel runs and
    ``shuffle_seed`` makes the (otherwise unseeded) training shuffle identical on every rank."""

    def __init__(self, idx, seed=False, rank=0, world=1, shuffle_seed=None):
        self.idx, self.seed, self.rank, self.world, self.shuffle_seed = idx, seed, rank, world, shuffle_seed
        self.epoch = 0

    def __iter__(self):
        order = list(self.idx.copy()) if not isinstance(self.idx, list) else self.idx.copy()
        if self.seed:
            np.random.seed(0)
            np.random.shuffle(order)
        elif self.shuffle_seed is not None:
            random.Random(self.shuffle_seed + self.epoch).shuffle(order)
        else:
            random.shuffle(order)
        self.epoch += 1
        if self.world > 1:
            n = len(order) // self.world * self.world if len(order) >= self.world else len(order)
            order = order[:n][self.rank::self.world] if n >= self.world else order
        yield from order

    def __len__(self):
        n = len(self.idx)
        return n // self.world if self.world > 1 and n >= self.world else n


# --------------------------------------------------------------------------------------- datasets
class ArrayDataset(Dataset):
    """In-memory HR stacks (uint8 [N, C, H, W]) with the attribute protocol the drivers consume
    (``val_idx``, ``extra_hr_files``, ``crop_res``, ``lr_scale``, ``is_lr``, ``hr_res``, ``n_frames``, ``_get_name``)."""

    def __init__(self, images, hr_res=512, lr_scale=4, crappifier=Poisson(), val_split=0.1, rotation=True, split_seed=0,
                 transforms=None, names=None, n_frames=-1):
        images = np.asarray(images)
        if images.ndim == 3:
            images = images[:, None]
        if images.dtype != np.uint8:
            raise ValueError("ArrayDataset expects uint8 images")
        self.images = images
        lr_scale = None if lr_scale == -1 else lr_scale
        self.n_frames = _get_n_frames(n_frames)
        self.slices = [1] * len(images)
        max_size = max(images.shape[-2:])
        self.val_idx = _get_val_idx(self.slices, val_split, split_seed)
        self.crop_res = min(hr_res, max_size)
        self.is_lr = lr_scale is None or max_size <= hr_res // lr_scale
        self.hr_res, self.lr_scale = hr_res, lr_scale if lr_scale is not None else 1
        self.crappifier, self.rotation, self.transforms = crappifier, rotation, transforms
        self.extra_hr_files = None
        self.compact = False        # True while train_paired feeds a captured graph from this dataset: uint8 items (see _tensor_ready)
        self.names = names if names is not None else [f"image{i}" for i in range(len(images))]

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx):
        if idx >= len(self):
            raise IndexError(f"Tried to retrieve invalid image. Index {idx} is not less than {len(self)} total image frame slices.")
        is_val = idx in self.val_idx
        rot = [bool(random.getrandbits(1)), random.choice((1, 2, (1, 2)))] if self.rotation and not is_val else False
        hr = self.images[idx]
        if self.is_lr:
            return _ready_lr(hr, self.hr_res // self.lr_scale, self.transforms, getattr(self, "compact", False))
        return _gen_pair(hr, self.hr_res, self.lr_scale, rot, self.crappifier, self.transforms, self.n_frames, getattr(self, "compact", False))

    def _get_name(self, idx):
        return self.names[idx]


class ImageDataset(ArrayDataset):
    """Folder of pre-tiled single-frame images (anything Pillow opens), reference arguments
    (pssr/data.py:13).  Multi-frame stacks, czi sheets and ``extra_path`` are outside this build's scope."""

    def __init__(self, path, hr_res=512, lr_scale=4, crappifier=Poisson(), n_frames=-1, extension="tif", val_split=0.1,
                 rotation=True, split_seed=0, extra_path=None, extra_scale=1, transforms=None):
        from PIL import Image
        self.path = Path(path) if type(path) is str else path
        if not path or not self.path.exists():
            raise FileNotFoundError(f'Path "{self.path}" does not exist.')
        files = sorted(f.split(str(self.path), maxsplit=1)[-1].strip("/") for f in glob.glob(f"{self.path}/**/*.{extension}", recursive=True))
        if not files:
            raise FileNotFoundError(f'No .{extension} files exist in path "{self.path}".')
        if extra_path is not None:
            raise NotImplementedError("extra_path is not supported by pssr2_amd.ImageDataset")
        stacks = []
        for f in files:
            im = Image.open(Path(self.path, f))
            frames = []
            for k in range(getattr(im, "n_frames", 1)):
                im.seek(k)
                frames.append(np.asarray(im.convert("L"), dtype=np.uint8))
            stacks.append(np.stack(frames))
        shapes = {s.shape for s in stacks}
        if len(shapes) != 1:
            raise ValueError("pssr2_amd.ImageDataset needs equally sized images")
        super().__init__(np.stack(stacks), hr_res, lr_scale, crappifier, val_split, rotation, split_seed, transforms,
                         [f.split(".")[0] for f in files], n_frames)
        self.hr_files = files


class SlidingArrayDataset(Dataset):
    """LR-mode sliding window over in-memory sheets (pssr/data.py:132-266 with ``lr_scale=-1``): tiles are
    row-major, ``stride = hr_res - overlap``, trailing remainders are dropped."""

    def __init__(self, sheets, hr_res=128, overlap=32, names=None, transforms=None):
        self.sheets = [np.asarray(s if np.asarray(s).ndim == 3 else np.asarray(s)[None]) for s in sheets]
        self.hr_res, self.lr_scale, self.stride = hr_res, 1, hr_res - overlap
        self.tiles = [int(np.prod(_n_tiles(s, hr_res, self.stride))) for s in self.sheets]
        self.val_idx = list(range(sum(self.tiles)))
        self.crop_res, self.is_lr, self.extra_hr_files, self.n_frames = hr_res, True, None, None
        self.names = names if names is not None else [f"sheet{i}" for i in range(len(self.sheets))]
        self.transforms = transforms
        self.compact = False        # see ArrayDataset.compact

    def __len__(self):
        return sum(self.tiles)

    def _locate(self, idx):
        for s, t in enumerate(self.tiles):
            if idx < t:
                return s, idx
            idx -= t
        raise IndexError(idx)

    def __getitem__(self, idx):
        s, t = self._locate(idx)
        return _tensor_ready(_sliding_tile(self.sheets[s], self.hr_res, self.stride, t), self.transforms, getattr(self, "compact", False))

    def _get_name(self, idx):
        s, t = self._locate(idx)
        return f"{self.names[s]}_{t}_0"


def synthetic_em_tile(index, res=512, channels=1):
    """Seeded synthetic EM-like uint8 tile: band-limited noise + white noise."""
    rng = np.random.default_rng(1234 + index)
    out = []
    for _ in range(channels):
        white = rng.standard_normal((res, res))
        f = np.fft.rfft2(white)
        ky, kx = np.fft.fftfreq(res)[:, None], np.fft.rfftfreq(res)[None, :]
        smooth = np.fft.irfft2(f * np.exp(-2 * (np.pi * 3.0) ** 2 * (kx ** 2 + ky ** 2)), s=(res, res))
        smooth /= smooth.std() + 1e-12
        out.append(np.clip(128 + 48 * smooth + 8 * rng.standard_normal((res, res)), 0, 255).astype(np.uint8))
    return np.stack(out)


# --------------------------------------------------------------------------------------- device path
class DevicePairGenerator:
    """(HR, LR) batches from uint8 HR tiles already resident in HBM, entirely with HIP kernels."""

    def __init__(self, lr_scale=4, crappifier=Poisson(), seed=0, tile_counter=None):
        self.lr_scale, self.crappifier, self.seed = lr_scale, crappifier, seed
        self.tile_counter = tile_counter      # optional device uint64 added to tile_offset in-kernel (hipGraph replay)

    def _stage(self, x, spec, seed, tile_offset, flags):
        from . import ops
        kind, intensity, gain, spread = spec
        if kind == "gaussian":
            return ops.crappify_gaussian(x, intensity, gain, spread, seed, tile_offset, flags, tile_counter=self.tile_counter)
        if kind == "poisson":
            return ops.crappify_poisson(x, intensity, gain, spread, seed, tile_offset, flags, tile_counter=self.tile_counter)
        if kind == "blur":
            if spread > 0:      # per-tile sigma from the tile's Philox stream
                return ops.gaussian_blur_tiles(x, intensity, spread, gain, seed, tile_offset, flags, tile_counter=self.tile_counter)
            return ops.gaussian_blur(x, intensity, gain, flags)
        if kind == "saltpepper":
            return ops.crappify_saltpepper(x, intensity, gain, spread, seed, tile_offset, flags, tile_counter=self.tile_counter)
        raise NotImplementedError(kind)

    def __call__(self, hr_u8: torch.Tensor, tile_offset: int = 0):
        """hr_u8: uint8 [B, C, H, W] on the device.  Returns float32 (hr, lr) like ``_gen_pair``."""
        from . import ops
        if not hr_u8.is_cuda or hr_u8.dtype != torch.uint8:
            raise RuntimeError("DevicePairGenerator needs a uint8 tensor on the MI355X")
        hr_u8 = hr_u8.contiguous()
        b, c, h, w = hr_u8.shape
        lr = ops.u8_to_f32(ops.bilinear_down_u8(hr_u8, h // self.lr_scale, w // self.lr_scale))
        cr = self.crappifier
        if cr is not None:
            spec = cr.device_spec()
            if isinstance(spec, list):
                clip = ops.CLIP if spec[0][0] == "clip" else 0
                stages = spec[1:]
                for i, st in enumerate(stages):
                    last = i == len(stages) - 1
                    lr = self._stage(lr, st, self.seed + 7919 * i, tile_offset, ops.ROUND_CLIP if last and clip else (clip if not last else 0))
                if not clip:
                    lr = self._stage(lr, ("gaussian", 0.0, 0.0, 0.0), 0, 0, ops.ROUND_CLIP)
            else:
                lr = self._stage(lr, spec, self.seed, tile_offset, ops.ROUND_CLIP)
        return ops.u8_to_f32(hr_u8), lr

    def from_stacks(self, stacks, hr_res, rotations=None, tile_offset: int = 0):
        """On-device ``_gen_pair`` (pssr/data.py:471-495) for a batch of uint8 stacks [C, H, W] resident in HBM: centred
        crop / reflect pad to ``hr_res``, rot90 / flip with the (host-drawn, reference-order) ``rotations``, then the
        Pillow-exact reduction and the crappifier.  Returns float32 (hr, lr)."""
        from . import ops
        rotations = [False] * len(stacks) if rotations is None else rotations
        return self(ops.gen_pair_geometry_u8(stacks, rotations, hr_res), tile_offset)


class DeviceTileDataset(Dataset):
    """``ArrayDataset`` whose uint8 HR stacks live in HBM: same constructor arguments, same attribute protocol
    (``val_idx``, ``extra_hr_files``, ``crop_res``, ``lr_scale``, ``is_lr``, ``hr_res``, ``n_frames``, ``_get_name``) and the same
    ``__getitem__`` contract (float32 CHW tensors, here already on the device), so ``train_paired`` / ``predict_images`` /
    ``test_metrics`` take it like any dataset.  In addition it can produce whole batches without touching the host
    (``draw_items`` + ``device_batch``): ``_gen_pair``'s crop / reflect pad / rot90 / flip (host-drawn in the reference's order,
    applied by one gather kernel), the Pillow-exact reduction and the crappifier (device Philox streams) as HIP launches whose
    only per-step inputs are device tensors -- which is what lets ``train_paired`` replay a whole training step as one hipGraph
    (pssr2_amd/fastpath.py).  Noise comes from the device generator: statistically, not bitwise, the numpy stream of the host path."""

    def __init__(self, images, hr_res=512, lr_scale=4, crappifier=Poisson(), val_split=0.1, rotation=True, split_seed=0,
                 transforms=None, names=None, n_frames=-1, device="cuda", seed=0):
        if transforms is not None:
            raise NotImplementedError("DeviceTileDataset applies no host transforms")
        images = torch.as_tensor(np.asarray(images) if not torch.is_tensor(images) else images)
        if images.dim() == 3:
            images = images[:, None]
        if images.dtype != torch.uint8:
            raise ValueError("DeviceTileDataset expects uint8 images")
        self.images = images.to(device).contiguous()
        lr_scale = None if lr_scale == -1 else lr_scale
        self.n_frames = _get_n_frames(n_frames)
        if self.n_frames is not None and self.n_frames[0] != self.n_frames[1]:
            raise NotImplementedError("DeviceTileDataset: 2.5-D frame slicing (n_frames=[lr, hr]) stays on the host path (ArrayDataset)")
        n = len(self.images)
        max_size = max(self.images.shape[-2:])
        self.val_idx = _get_val_idx([1] * n, val_split, split_seed)
        self.crop_res = min(hr_res, max_size)
        self.is_lr = lr_scale is None or max_size <= hr_res // lr_scale
        self.hr_res, self.lr_scale = hr_res, lr_scale if lr_scale is not None else 1
        self.crappifier, self.rotation, self.transforms = crappifier, rotation, None
        self.extra_hr_files = None
        self.names = names if names is not None else [f"image{i}" for i in range(n)]
        self._val_set, self._val_key = set(self.val_idx), None
        self.tile_counter = torch.zeros(1, dtype=torch.int64, device=self.images.device)
        self.gen = DevicePairGenerator(self.lr_scale, crappifier, seed=seed, tile_counter=self.tile_counter)
        self._item_bytes = 24             # struct pssr_gather_item {src, sh, sw, rot, flip_axis}

    def __len__(self):
        return len(self.images)

    def _get_name(self, idx):
        return self.names[idx]

    def _draw_rotation(self, idx):
        # ``idx in self.val_idx`` as upstream (pssr/data.py:103), with the list hashed once per assignment / length change: users enlarge
        # val_idx after training to predict every image
        key = (id(self.val_idx), len(self.val_idx))
        if key != self._val_key:
            self._val_set, self._val_key = set(self.val_idx), key
        if self.rotation and idx not in self._val_set:
            return [bool(random.getrandbits(1)), random.choice((1, 2, (1, 2)))]      # the reference's draws, in its order
        return False

    def draw_items(self, indices):
        """Gather table (int64 [n, 3] on the device = n ``pssr_gather_item``) for these dataset indices, drawing the training
        rotations exactly as ``__getitem__`` would for the same sequence of indices."""
        import struct
        c, h, w = self.images.shape[1:]
        base, stride = self.images.data_ptr(), c * h * w
        buf = bytearray()
        for i in indices:
            rot = self._draw_rotation(int(i))
            axis = -1
            if rot:
                axis = 3 if isinstance(rot[1], (tuple, list)) else int(rot[1])
            buf += struct.pack("<Qiiii", base + int(i) * stride, h, w, int(bool(rot and rot[0])), axis)
        if not buf:                        # an empty order (val_split = 0, a rank without validation items): torch.frombuffer rejects b""
            return torch.zeros(0, 3, dtype=torch.int64, device=self.images.device)
        return torch.frombuffer(buf, dtype=torch.int64).view(-1, 3).to(self.images.device)

    def device_batch(self, items):
        """items: int64 [b, 3] device rows of ``draw_items``.  Returns float32 (hr, lr) on the device, or lr alone in LR mode.  No
        host synchronisation, no host-side data: capturable in a hipGraph (the Philox tile counter advances on the device)."""
        from . import _lib as L, ops
        b, c = items.shape[0], self.images.shape[1]
        res = self.hr_res // self.lr_scale if self.is_lr else self.hr_res
        out = torch.empty(b, c, res, res, dtype=torch.uint8, device=self.images.device)
        L.check(L.lib().pssr_gen_pair_geometry_u8(L.ptr(items), b, L.ptr(out), c, res, L.stream_ptr()), "pssr_gen_pair_geometry_u8")
        if self.is_lr:
            return ops.u8_to_f32(out)
        hr, lr = self.gen(out)
        ops.counter_add(self.tile_counter, b)
        return hr, lr

    def __getitem__(self, idx):
        if idx >= len(self):
            raise IndexError(f"Tried to retrieve invalid image. Index {idx} is not less than {len(self)} total image frame slices.")
        out = self.device_batch(self.draw_items([idx]))
        return out[0] if self.is_lr else (out[0][0], out[1][0])
