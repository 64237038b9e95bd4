"""``predict_images`` with the reference's signature (pssr/predict.py:11-83): batched no-grad forward,
clip + uint8 truncation on the device (csrc/elementwise.hip: pssr_clip_u8), crop, dict or file output."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, Dataset

from . import distributed as D
from . import ops
from .data import _slice_center
from .util import _get_callbacks

try:
    from tqdm import tqdm
except ImportError:                                   # pragma: no cover
    def tqdm(it, **kw):
        return it


def _pred_array(data: torch.Tensor, n_frames: int = 1):
    """``np.clip(x, 0, 255).astype(np.uint8)`` (truncation) + centre-frame slice (pssr/predict.py:245-246)."""
    if not data.is_cuda:
        raise RuntimeError("pssr2_amd.predict runs on an MI355X (HIP) device only; there is no CPU fallback")
    x = data.detach().contiguous().float()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    ops.clip_u8(x, out)
    return _slice_center(out.cpu().numpy(), n_frames)


def _pred_tensor(data: torch.Tensor, n_frames: int = 1):
    """``_pred_array`` that leaves the uint8 result in HBM (for the device-side normalisation and metrics)."""
    if not data.is_cuda:
        raise RuntimeError("pssr2_amd.predict runs on an MI355X (HIP) device only; there is no CPU fallback")
    x = data.detach().contiguous().float()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    ops.clip_u8(x, out)
    return _slice_center(out, n_frames)


def predict_images(model: nn.Module, dataset: Dataset, device: str = "cpu", batch_size=None, out_dir: str = "preds", norm: bool = False,
                   prefix: str = None, dataloader_kwargs=None, callbacks=None):
    r"""Predicts high-resolution images for ``dataset.val_idx``; returns ``{name: uint8 [C,H,W]}`` when
    ``out_dir`` is None, else writes ``{out_dir}/{prefix_}{name}.tif`` through Pillow."""
    dataloader_kwargs = {} if dataloader_kwargs is None else dataloader_kwargs
    batch_size = 1 if batch_size is None else batch_size
    if norm and dataset.is_lr:
        raise ValueError("Dataset must be paired with high-low-resolution images for normalization.")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    callbacks, callback_locals = _get_callbacks(callbacks)
    rank, world = D.rank_world()

    model.to(device)
    model.eval()
    idx = list(dataset.val_idx)
    if world > 1:                                     # contiguous chunk per rank keeps output naming intact
        per = (len(idx) + world - 1) // world
        first = rank * per
        idx = idx[first:first + per]
    else:
        first = 0
    # device-made batches replayed as a hipGraph when the dataset offers them (pssr2_amd/fastpath.py); ``dataset.device_outputs``
    # (an extension of such datasets) keeps the uint8 predictions in HBM: the dict then holds device tensors
    from . import fastpath
    fast = fastpath.supports(model, dataset, device) and not dataloader_kwargs and len(idx) > 0
    host_fast = not fast and fastpath.supports_host(model, dataset, device) and len(idx) > 0
    keep_dev = bool(getattr(dataset, "device_outputs", False)) and not out_dir and not norm
    compact = False
    if host_fast:
        # host batches (any dataset through a DataLoader): one captured forward over a static input buffer per (dataset, batch size)
        cache = model._engine.__dict__.setdefault("_eval_steppers", {})
        evaler = cache.get((id(dataset), batch_size, "host"))
        if evaler is None or evaler.dataset is not dataset:
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            evaler = cache[(id(dataset), batch_size, "host")] = fastpath.EvalStepper(model, dataset, batch_size, device, to_u8=True, weights_move=False,
                                                                                     host=True)
        evaler.begin()
        # a dataset of this package hands the replay uint8 items (pssr2_amd/data.py: _tensor_ready); taken back when the loop below ends
        compact = getattr(dataset, "compact", None) is False and os.environ.get("PSSR_HOST_COMPACT", "1") != "0"
        if compact:
            dataset.compact = True
        dataloader = DataLoader(dataset, batch_size, sampler=idx, **dataloader_kwargs)
    elif fast:
        # one stepper (and one captured graph) per (dataset, batch size): a second call over the same dataset only replays
        cache = model._engine.__dict__.setdefault("_eval_steppers", {})
        evaler = cache.get((id(dataset), batch_size))
        if evaler is None or evaler.dataset is not dataset or len(idx) > evaler.cur.table.shape[0]:
            if len(cache) >= 4:           # each stepper pins a captured graph and its memory pool: keep the most recent few
                cache.pop(next(iter(cache)))
            evaler = cache[(id(dataset), batch_size)] = fastpath.EvalStepper(model, dataset, batch_size, device, to_u8=True, weights_move=False)
        dataloader = range(evaler.begin(idx))
    else:
        dataloader = DataLoader(dataset, batch_size, sampler=idx, **dataloader_kwargs)
    outs, cur_idx = {}, first
    from .train import _restore_compact
    with torch.no_grad(), _restore_compact(dataset, host_fast and compact):
        for item in tqdm(dataloader, disable=rank != 0):
            if fast or host_fast:
                hr_dev, lr, _, _, u8 = evaler.step() if fast else evaler.step((item,) if dataset.is_lr else tuple(item))
                hr_hat = u8.clone() if keep_dev else _slice_center(u8.cpu().numpy(), 1)      # u8 is the graph's static output buffer
            else:
                lr = item if dataset.is_lr else item[1]
                lr = lr.to(device)
                hr_hat = _pred_array(model(lr))
            if norm:      # pssr/predict.py:63-64: intensities matched to the ground truth of the paired dataset
                from .util import normalize_preds
                _, hr_hat = normalize_preds(_pred_array(hr_dev if (fast or host_fast) else item[0].to(device)), hr_hat)
            crop_res = dataset.crop_res if not dataset.is_lr else dataset.crop_res * (hr_hat.shape[-1] // lr.shape[-1])
            hr_hat = hr_hat[:, :, :crop_res, :crop_res]
            for batch_idx, image_idx in enumerate(range(cur_idx, min(cur_idx + batch_size, first + len(idx)))):
                name = dataset._get_name(image_idx)
                if out_dir:
                    from PIL import Image
                    frames = [Image.fromarray(f) for f in hr_hat[batch_idx]]
                    frames[0].save(f"{out_dir}/{prefix + '_' if prefix else ''}{name}.tif", save_all=len(frames) > 1, append_images=frames[1:])
                else:
                    outs[name] = hr_hat[batch_idx]
                for i, callback in enumerate(callbacks):
                    callback(locals()) if callback_locals[i] else callback()
            cur_idx += batch_size
    if out_dir is None:
        return outs


def predict_sheet(model: nn.Module, sheet, tile_res: int = 128, overlap: int = 32, margin: int = 0, batch_size: int = 128,
                  device: str = "cuda", to_numpy: bool = True):
    r"""Whole-sheet prediction entirely on the device: what ``predict_images`` on a ``SlidingDataset`` in LR mode
    (pssr/data.py:132-266, pssr/predict.py:11-83) followed by ``reassemble_sheets`` (pssr/util.py:54-137) computes, without
    the per-tile host round trips.  ``sheet``: uint8 array or tensor [C, H, W] (or [H, W]) of the low-resolution image.
    Tiles of ``tile_res`` LR pixels with ``overlap`` are cut on the device, predicted in batches, truncated to uint8
    (``_pred_array``), and stitched with overlap averaging and ``margin`` trimming (in LR pixels times the model's scale,
    as upstream).  Returns uint8 [C_out, H', W'] with H' = (n_rows*(tile_res-overlap)+overlap)*scale."""
    if margin > overlap:
        raise ValueError(f"The value of margin cannot be greater than overlap. Given {margin} and {overlap} respectively.")
    sheet = torch.as_tensor(np.asarray(sheet) if not torch.is_tensor(sheet) else sheet)
    if sheet.dim() == 2:
        sheet = sheet[None]
    if sheet.dtype != torch.uint8:
        raise ValueError("predict_sheet expects a uint8 sheet (the reference's uint8 tif path)")
    sheet = sheet.to(device).contiguous()
    c, h, w = sheet.shape
    if h < tile_res or w < tile_res:
        raise ValueError(f"sheet {h}x{w} is smaller than one tile ({tile_res})")
    stride = tile_res - overlap
    n_rows, n_cols = (h - tile_res) // stride + 1, (w - tile_res) // stride + 1
    n_tiles = n_rows * n_cols
    model.to(device)
    model.eval()
    preds = None
    with torch.no_grad():
        for t0 in range(0, n_tiles, batch_size):
            nt = min(batch_size, n_tiles - t0)
            lr = ops.sliding_tiles_u8(sheet, tile_res, stride, t0, nt)
            y = model(lr).contiguous().float()
            if preds is None:
                preds = torch.empty(n_tiles, *y.shape[1:], dtype=torch.uint8, device=y.device)
            ops.clip_u8(y, preds[t0:t0 + nt])
    scale = preds.shape[-1] // tile_res
    # upstream passes overlap*lr_scale and the raw margin to _patch_images (pssr/util.py:99)
    out = ops.patch_tiles_u8(preds, n_rows, n_cols, overlap * scale, margin)
    return out.cpu().numpy() if to_numpy else out


def test_metrics(model: nn.Module, dataset: Dataset, device: str = "cpu", metrics=("mse", "pixel", "psnr", "ssim"), avg: bool = True,
                 norm: bool = True, callbacks=None):
    r"""Image restoration metrics of predicted vs ground truth images over ``dataset.val_idx`` (pssr/predict.py:144-211): same
    arguments and return value.  Prediction, uint8 cast, (``norm``) intensity normalisation and the per-image statistics all run on
    the MI355X (csrc/metrics.hip): the squared-difference sum is an exact integer and ``ssim`` is scikit-image's
    ``structural_similarity(data_range=255)`` (uniform 7x7 windows, sample covariance, interior mean) evaluated from exact integer
    window sums; only two doubles per image come back to the host.  Like the reference (pssr/predict.py:180) every iteration
    evaluates ``dataset[0]``."""
    import math
    from .util import pixel_metric
    callbacks, callback_locals = _get_callbacks(callbacks)
    image_range = 255
    metrics = [metrics] if type(metrics) is str else list(metrics)
    out = {m: [] for m in metrics}
    model.to(device)
    model.eval()

    with torch.no_grad():
        for _ in tqdm(dataset.val_idx):
            hr, lr = dataset[0]
            hr, lr = hr.to(device).unsqueeze(0), lr.to(device).unsqueeze(0)
            hr_hat = model(lr)
            hr_dev, hat_dev = _pred_tensor(hr), _pred_tensor(hr_hat)
            crop_res = dataset.crop_res if not dataset.is_lr else dataset.crop_res * (hr_hat.shape[-1] // lr.shape[-1])
            hr_dev, hat_dev = hr_dev[:, :, :crop_res, :crop_res].contiguous(), hat_dev[:, :, :crop_res, :crop_res].contiguous()
            if norm:
                hr_dev, hat_dev = ops.normalize_preds_u8(hr_dev, hat_dev)
            n, c, h, w = hr_dev.shape
            if "ssim" in out and c > 1:
                raise ValueError("ssim: multi-channel images form a volume thinner than the 7x7x7 window (scikit-image raises here too)")
            stats = ops.image_metrics_u8(hr_dev, hat_dev).reshape(n, c, 2).cpu()
            for i in range(n):
                err = float(stats[i, :, 0].sum()) / (c * h * w)           # mean squared error at uint8 scale (exact sum)
                mse = err / image_range ** 2
                if "mse" in out:
                    out["mse"].append(mse)
                if "pixel" in out:
                    out["pixel"].append(pixel_metric(mse, image_range))
                if "psnr" in out:
                    out["psnr"].append(10 * math.log10(image_range ** 2 / err) if err > 0 else float("inf"))
                if "ssim" in out:
                    out["ssim"].append(float(stats[i, 0, 1]))
            if any(callback_locals):
                hr, hr_hat = hr_dev.cpu().numpy(), hat_dev.cpu().numpy()       # the arrays the reference's callbacks see
            for i, callback in enumerate(callbacks):
                callback(locals()) if callback_locals[i] else callback()
    return {m: (sum(v) / len(v) if avg else v) for m, v in out.items()}


test_metrics.__test__ = False      # a library function, not a pytest test (the reference's conftest deselects it the same way)
