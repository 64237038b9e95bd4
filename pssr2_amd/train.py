"""``train_paired`` with the reference's signature, return value and loop semantics
(pssr/train.py:19-166), running the model / loss / optimizer kernels of libpssr_mi355.so.

Additions with no reference counterpart (documented in DESIGN.md): when ``torch.distributed`` is
initialised the training indices are sharded per rank with a rank-identical shuffle, gradients are
mean-all-reduced (overlapped with backward through ``Engine.attach_reducer``), validation loss is
averaged over ranks, and only rank 0 writes checkpoints / collages.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
from torch.utils.data import DataLoader, Dataset

from . import distributed as D
from .data import _invert_idx, _RandomIterIdx
from .util import _get_callbacks, _psnr_metric, pixel_metric

try:
    from tqdm import tqdm
except ImportError:                                   # pragma: no cover
    def tqdm(it, **kw):
        return it


def _metric_ssim(hr_hat, hr, image_range):
    from .util import ssim
    try:
        return float(ssim(hr_hat.detach(), hr, data_range=image_range))
    except Exception:
        return float("nan")


def _collage(lr, hr_hat, hr, crop_res, lr_scale):
    """Small PIL collage of (LR | prediction | HR) rows for ``collage_dir`` (pssr/train.py:155-158)."""
    import numpy as np
    from PIL import Image
    rows = []
    for i in range(min(len(lr), 4)):
        up = np.kron(np.clip(lr[i, lr.shape[1] // 2].numpy(), 0, 255), np.ones((lr_scale, lr_scale)))
        tiles = [up, np.clip(hr_hat[i, hr_hat.shape[1] // 2].numpy(), 0, 255), np.clip(hr[i, hr.shape[1] // 2].numpy(), 0, 255)]
        rows.append(np.concatenate([t[:crop_res, :crop_res] for t in tiles], axis=1))
    return Image.fromarray(np.concatenate(rows, axis=0).astype(np.uint8))


class _restore_compact:
    """Takes the dataset out of compact (uint8 item) mode when the driver leaves, however it leaves."""

    def __init__(self, dataset, active):
        self.dataset, self.active = dataset, active

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.active:
            self.dataset.compact = False
        return False


def train_paired(model: nn.Module, dataset: Dataset, batch_size: int, loss_fn: nn.Module, optim: torch.optim.Optimizer, epochs: int,
                 device: str = "cpu", scheduler=None, log_frequency: int = 50, checkpoint_dir: str = None, collage_dir: str = None,
                 clamp: bool = False, dataloader_kwargs=None, callbacks=None):
    r"""Trains ``model`` on paired high-/low-resolution data; returns ``(train_losses, val_losses)``."""
    dataloader_kwargs = {} if dataloader_kwargs is None else dataloader_kwargs
    callbacks, callback_locals = _get_callbacks(callbacks)
    image_range = 255
    rank, world = D.rank_world()

    train_idx = _invert_idx(dataset.val_idx, len(dataset))
    train_sampler = _RandomIterIdx(train_idx, rank=rank, world=world, shuffle_seed=0 if world > 1 else None)
    val_sampler = _RandomIterIdx(dataset.val_idx, seed=True, rank=rank, world=world)
    include_metric = type(scheduler) == torch.optim.lr_scheduler.ReduceLROnPlateau

    model.to(device)
    engine = getattr(model, "_engine", None)
    # hipGraph replay of whole steps when the dataset makes its batches on the device (pssr2_amd/fastpath.py); everything else
    # (host datasets, DataLoader workers, user crappifier subclasses, ``extra`` losses) takes the reference's loop below
    from . import fastpath
    fast = fastpath.supports(model, dataset, device) and not dataloader_kwargs
    # host datasets (the reference's own ImageDataset / SlidingDataset through a DataLoader, workers and all): the same replay over
    # static input buffers, fed by one asynchronous host-to-device copy per batch
    host_fast = not fast and fastpath.supports_host(model, dataset, device)
    # a dataset of this package feeding the replay hands over uint8 items (every value is an integer in [0, 255]); the feed converts on the
    # device.  The flag travels to the DataLoader workers with the dataset and is taken back when the driver returns
    compact = host_fast and getattr(dataset, "compact", None) is False and os.environ.get("PSSR_HOST_COMPACT", "1") != "0"
    if compact:
        dataset.compact = True
    if fast:
        train_dataloader = val_dataloader = None
    else:
        train_dataloader = DataLoader(dataset, batch_size, sampler=train_sampler, **dataloader_kwargs)
        val_dataloader = DataLoader(dataset, batch_size, sampler=val_sampler, **dataloader_kwargs)
    if world > 1:
        D.broadcast_module(model)
        if engine is not None and engine.reducer is None and not fast and not host_fast:
            engine.attach_reducer()

    # fp16 storage (model.compute_dtype = torch.float16) needs loss scaling; f32 / bf16 do not
    scaler = None
    if getattr(model, "compute_dtype", None) == torch.float16:
        from .optim import LossScaler
        scaler = LossScaler()

    stepper = evaler = None
    if fast:
        stepper = fastpath.TrainStepper(model, dataset, batch_size, loss_fn, optim, clamp, image_range, scaler, len(train_sampler), device)
        evaler = fastpath.EvalStepper(model, dataset, batch_size, device, loss_fn=loss_fn, clamp=clamp, image_range=image_range)
    elif host_fast:
        stepper = fastpath.TrainStepper(model, dataset, batch_size, loss_fn, optim, clamp, image_range, scaler, 0, device, host=True)
        evaler = fastpath.EvalStepper(model, dataset, batch_size, device, loss_fn=loss_fn, clamp=clamp, image_range=image_range, host=True)

    train_losses, val_losses = [], []
    # world > 1: an exception on one rank (a callback's, say) ends every rank within seconds (pssr2_amd/distributed.py: failure_watch)
    with _restore_compact(dataset, compact), D.failure_watch("train_paired"):
        for epoch in range(epochs):
            model.train()
            if rank == 0:
                print(f"Epoch {epoch}:")
            if fast:
                progress = tqdm(range(stepper.begin_epoch(list(train_sampler))), disable=rank != 0)
            else:
                progress = tqdm(train_dataloader, disable=rank != 0)
            for batch_idx, data in enumerate(progress):
                if fast:
                    hr, lr, hr_hat, loss = stepper.step()
                elif host_fast:
                    hr, lr, hr_hat, loss = stepper.step(data)
                else:
                    if dataset.extra_hr_files is None:
                        hr, lr = data
                    else:
                        (hr, lr), extra = data
                        extra = extra.to(device)
                    hr, lr = hr.to(device), lr.to(device)

                    hr_hat = model(lr)
                    if clamp:
                        hr_hat = torch.clamp(hr_hat, 0, image_range)
                    loss = loss_fn(hr_hat / image_range, hr / image_range) if dataset.extra_hr_files is None \
                        else loss_fn(hr_hat / image_range, hr / image_range, extra / image_range)
                    (scaler.scale(loss) if scaler is not None else loss).backward()
                    if world > 1 and engine is None:
                        D.allreduce_mean_([p.grad for p in model.parameters() if p.grad is not None])
                    if scaler is not None:
                        scaler.step(optim, list(model.parameters()))
                    else:
                        optim.step()
                    optim.zero_grad()

                if batch_idx % log_frequency == 0 or batch_idx == len(progress) - 1:
                    train_losses.append(loss.item())
                    mse = nn.functional.mse_loss(hr_hat.detach() / image_range, hr / image_range)
                    if rank == 0 and hasattr(progress, "set_description"):
                        progress.set_description(f"pixel[{pixel_metric(mse.item(), image_range):.2f}], psnr[{_psnr_metric(mse):.2f}], "
                                                 f"ssim[{_metric_ssim(hr_hat, hr, image_range):.3f}]")
                if batch_idx == max(len(progress), 2) - 2:
                    last_full = [lr.cpu(), hr_hat.detach().cpu(), hr.cpu()]       # accessible from callbacks via locals
                for idx, callback in enumerate(callbacks):
                    callback(locals()) if callback_locals[idx] else callback()
            if fast or host_fast:
                stepper.finish()

            model.eval()
            if rank == 0:
                print(f"Epoch {epoch} validation...")
            val_loss = []
            if fast:
                progress = tqdm(range(evaler.begin(list(val_sampler))), disable=rank != 0)
            else:
                if host_fast:
                    evaler.begin()
                progress = tqdm(val_dataloader, disable=rank != 0)
            with torch.no_grad():
                for batch_idx, data in enumerate(progress):
                    if fast:
                        hr, lr, hr_hat, loss, _ = evaler.step()
                    elif host_fast:
                        hr, lr, hr_hat, loss, _ = evaler.step(tuple(data))
                    else:
                        if dataset.extra_hr_files is None:
                            hr, lr = data
                        else:
                            (hr, lr), extra = data
                            extra = extra.to(device)
                        hr, lr = hr.to(device), lr.to(device)
                        hr_hat = model(lr)
                        if clamp:
                            hr_hat = torch.clamp(hr_hat, 0, image_range)
                        loss = loss_fn(hr_hat / image_range, hr / image_range) if dataset.extra_hr_files is None \
                            else loss_fn(hr_hat / image_range, hr / image_range, extra / image_range)
                        val_loss.append(loss.detach().float().reshape(1))            # stays on device: one sync per epoch
                    if batch_idx == max(len(progress), 2) - 2:
                        last_full_val = [lr.cpu(), hr_hat.cpu(), hr.cpu()]
            if fast or host_fast:
                stat = evaler.mean_loss_stat()
                engine.mark_weights_changed()
            else:
                stat = torch.stack([torch.cat(val_loss).sum(), torch.tensor(float(len(val_loss)), device=val_loss[0].device)]) \
                    if val_loss else torch.zeros(2, device=device)
            if world > 1:
                torch.distributed.all_reduce(stat)
            val_loss = (stat[0] / stat[1].clamp(min=1)).item()
            val_losses.append(val_loss)
            if rank == 0:
                print(f"Epoch {epoch} validation loss: {val_loss:4f}\n")

            if checkpoint_dir and epoch < epochs - 1 and rank == 0:
                os.makedirs(checkpoint_dir, exist_ok=True)
                torch.save(model.state_dict(), f"{checkpoint_dir}/checkpoint{epoch}_{model.__class__.__name__}_{val_loss:.4f}.pth")
            if collage_dir and rank == 0:
                os.makedirs(collage_dir, exist_ok=True)
                _collage(*last_full_val, crop_res=dataset.crop_res, lr_scale=dataset.lr_scale).save(f"{collage_dir}/epoch{epoch}_loss{val_loss:.4f}.png")
            if scheduler:
                scheduler.step(val_loss) if include_metric else scheduler.step()
                if hasattr(optim, "sync_device_lr"):
                    optim.sync_device_lr()

    return train_losses, val_losses
