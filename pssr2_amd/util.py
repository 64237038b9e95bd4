"""Loss, metrics and small host helpers with the reference's names (pssr/util.py).

``SSIMLoss`` keeps the reference signature and semantics (pssr/util.py:10-52) but runs as two HIP
kernel sweeps over the image pyramid (csrc/loss.hip) instead of ~60 torch ops; the third-party
pytorch_msssim algorithm it stands for is restated in oracle/loss_ref.py.
"""
from __future__ import annotations

import inspect
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)    # pytorch_msssim default level weights


def _gauss_1d(size: int, sigma: float):
    coords = torch.arange(size, dtype=torch.float) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return (g / g.sum()).tolist()


_CONST_CACHE = {}      # small per-shape device constants (kept out of hipGraph capture)


class _SSIMLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, cfg, in_div=1.0):
        """in_div != 1: the loss of (x / in_div, y / in_div) with the divisions done inside the training kernels (11-tap window, x needing a
        gradient); the returned gradient is wrt the undivided x."""
        if not x.is_cuda:
            raise RuntimeError("pssr2_amd.SSIMLoss runs on an MI355X (HIP) device only; there is no CPU fallback")
        win, mix, ms, k1, k2, data_range, lvl_w = cfg
        x = x.detach().contiguous().float()
        y = y.detach().contiguous().float()
        n, c, h, w = x.shape
        planes, k = n * c, len(win)
        levels = len(lvl_w) if ms else 1
        if ms and min(h, w) <= (k - 1) * 2 ** 4:
            raise AssertionError(f"Image size should be larger than {(k - 1) * 2 ** 4} due to the 4 downsamplings in ms-ssim")
        c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
        dev = x.device
        fused_div = in_div != 1.0 and k == 11 and ctx.needs_input_grad[0]
        if in_div != 1.0 and not fused_div:             # the plain kernels take the quotients as tensors
            x, y = x / in_div, y / in_div
        div0 = in_div if fused_div else 1.0
        xs, ys, dims = [x], [y], [(h, w)]
        for lv in range(1, levels):
            hh, ww = dims[-1]
            ho, wo = (hh + 2 * (hh & 1) - 2) // 2 + 1, (ww + 2 * (ww & 1) - 2) // 2 + 1
            xo = torch.empty(planes, ho, wo, device=dev)
            yo = torch.empty(planes, ho, wo, device=dev)
            ops.avgpool2_pair(xs[-1], ys[-1], xo, yo, planes, hh, ww, div0 if lv == 1 else 1.0)
            xs.append(xo), ys.append(yo), dims.append((ho, wo))
        # training with the default 11-tap window: keep the per-position derivatives for the backward pass (ops.ssim_level_fwd_adj);
        # that kernel spreads its sums over `stripes` copies, folded by msssim_weights_striped
        adjs = None
        if k == 11 and ctx.needs_input_grad[0]:
            adjs = [torch.empty(planes * 3 * hh * ww, device=dev) for hh, ww in dims]
        stripes = 16 if adjs is not None else 1
        sstride = levels * planes * 2 + 2
        rows = 2 * stripes if stripes > 1 else 1        # the striped kernel adds every sum as two exact pieces (order-independent)
        sums = torch.zeros(rows * sstride + (sstride if stripes > 1 else 0), dtype=torch.float64, device=dev)      # (+ the folded copy)
        l1_sum = sums[sstride - 2:sstride - 1] if mix < 1 else None
        for l in range(levels):
            hh, ww = dims[l]
            if adjs is not None:
                ops.ssim_level_fwd_adj(xs[l], ys[l], planes, hh, ww, win, c1, c2, l == levels - 1, sums[l * planes * 2:],
                                       l1_sum if l == 0 else None, stripes, sstride, adjs[l], div0 if l == 0 else 1.0)
            else:
                ops.ssim_level_fwd(xs[l], ys[l], planes, hh, ww, win, c1, c2, sums[l * planes * 2:], l1_sum if l == 0 else None)
        ckey = (tuple(dims), k, tuple(lvl_w) if ms else (1.0,), str(dev))
        cached = _CONST_CACHE.get(ckey)
        if cached is None:
            cached = _CONST_CACHE[ckey] = (
                torch.tensor([float((hh - k + 1) * (ww - k + 1)) for hh, ww in dims], dtype=torch.float64, device=dev),
                torch.tensor(list(lvl_w) if ms else [1.0], dtype=torch.float32, device=dev))
        nvalid, lw = cached
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        wts = torch.empty(levels * planes, dtype=torch.float32, device=dev)
        l1c = torch.empty(1, dtype=torch.float32, device=dev)
        if stripes > 1:
            ops.msssim_weights_striped(sums, stripes, sstride, sums[rows * sstride:], levels, planes, nvalid, lw, ms, mix, l1_sum, float(x.numel()),
                                       None, loss, wts, l1c)
            sums = sums[rows * sstride:]                          # the folded copy: what backward's weights pass reads
            l1_sum = sums[levels * planes * 2:levels * planes * 2 + 1] if l1_sum is not None else None
        else:
            ops.msssim_weights(sums, levels, planes, nvalid, lw, ms, mix, l1_sum, float(x.numel()), None, loss, wts, l1c)
        ctx.saved = (xs, ys, dims, sums, nvalid, lw, l1_sum, cfg, planes, x.shape, adjs, div0, in_div if not fused_div else 1.0)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        xs, ys, dims, sums, nvalid, lw, l1_sum, cfg, planes, shape, adjs, div0, post_div = ctx.saved
        win, mix, ms, k1, k2, data_range, lvl_w = cfg
        levels = len(xs)
        dev = xs[0].device
        c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
        go = grad_out.detach().float().reshape(1).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        wts = torch.empty(levels * planes, dtype=torch.float32, device=dev)
        l1c = torch.empty(1, dtype=torch.float32, device=dev)
        ops.msssim_weights(sums, levels, planes, nvalid, lw, ms, mix, l1_sum, float(xs[0].numel()), go, loss, wts, l1c)
        dcoarse, hc, wc = None, 0, 0
        for l in range(levels - 1, -1, -1):
            hh, ww = dims[l]
            dx = torch.empty(planes, hh, ww, device=dev)
            use_ssim = (l == levels - 1)
            l1_arg = l1c if (l == 0 and l1_sum is not None) else None
            if adjs is not None:
                ops.ssim_level_bwd_adj(xs[l], ys[l], adjs[l], planes, hh, ww, win, wts[l * planes:], dcoarse, hc, wc, l1_arg, dx,
                                       div0 if l == 0 else 1.0)
            else:
                ops.ssim_level_bwd(xs[l], ys[l], planes, hh, ww, win, c1, c2, wts[l * planes:], use_ssim, dcoarse, hc, wc, l1_arg, dx)
            dcoarse, hc, wc = dx, hh, ww
        g = dcoarse.view(shape)
        if post_div != 1.0:
            g = g / post_div
        return g, None, None, None


class SSIMLoss(nn.Module):
    def __init__(self, channels: int = 1, mix: float = .8, win_size: int = 11, win_sigma: float = 1.5, ms: bool = True, kwargs=None):
        r"""SSIM / MS-SSIM loss mixed with a Gaussian-weighted L1 term (Zhao et al., 2018), reference
        signature (pssr/util.py:11).  ``kwargs`` accepts the pytorch_msssim options ``K`` and
        ``weights``.  For ``channels > 1`` the L1 window is depthwise (the reference's [1,1,k,k]
        window raises in torch for C > 1).
        """
        super().__init__()
        kwargs = {} if kwargs is None else dict(kwargs)
        if win_size % 2 != 1:
            raise ValueError("Window size should be odd.")
        self.K = tuple(kwargs.pop("K", (0.01, 0.03)))
        self.weights = tuple(kwargs.pop("weights", None) or MS_WEIGHTS)
        if kwargs:
            raise TypeError(f"unsupported pytorch_msssim options on the MI355X path: {sorted(kwargs)}")
        self.win = _gauss_1d(win_size, win_sigma)
        self.channels, self.win_size, self.mix, self.ms = channels, win_size, mix, ms

    def forward(self, input, target):
        if input.shape != target.shape:
            raise ValueError(f"Input images should have the same dimensions, but got {input.shape} and {target.shape}.")
        cfg = (self.win, float(self.mix), bool(self.ms), self.K[0], self.K[1], 1.0, self.weights)
        return _SSIMLossFunction.apply(input, target, cfg)

    def forward_divided(self, input, target, divisor: float):
        """``self(input / divisor, target / divisor)`` -- what pssr/train.py:101 computes with divisor 255 -- without the two quotient
        tensors and the gradient's division pass: the training kernels scale on load (a rounded multiplication by the f32 reciprocal 1/in_div, which is how torch divides a device tensor by a scalar: same values)."""
        if input.shape != target.shape:
            raise ValueError(f"Input images should have the same dimensions, but got {input.shape} and {target.shape}.")
        cfg = (self.win, float(self.mix), bool(self.ms), self.K[0], self.K[1], 1.0, self.weights)
        return _SSIMLossFunction.apply(input, target, cfg, float(divisor))


def ssim(X, Y, data_range=255, win_size=11, win_sigma=1.5):
    """Mean SSIM (the in-loop metric of pssr/train.py:109), forward only."""
    cfg = (_gauss_1d(win_size, win_sigma), 1.0, False, 0.01, 0.03, float(data_range), (1.0,))
    with torch.no_grad():
        return 1 - _SSIMLossFunction.apply(X, Y, cfg)


def pixel_metric(mse: float, image_range: int = 255):
    r"""Average pixel error from a mean squared error (pssr/util.py:207-215)."""
    return math.sqrt(mse) * image_range


def _psnr_metric(mse):
    return 20 * torch.log10(1 / torch.sqrt(mse))


def _force_list(item):
    if type(item) is not list:
        try:
            return list(item)
        except TypeError:
            return [item]
    return item


def _get_callbacks(raw):
    """1-argument callbacks receive ``locals()`` of the driver loop (pssr/util.py:228-231)."""
    callbacks = [] if raw is None else _force_list(raw)
    takes_locals = [len([a for a in inspect.getfullargspec(cb).args if a != "self"]) == 1 for cb in callbacks]
    return callbacks, takes_locals


def _patch_images(batched, n_cols, n_rows, overlap, margin):
    """Overlap-averaged stitching of row-major tiles (pssr/util.py:116-137), host numpy."""
    size = batched.shape[-1]
    step = size - overlap
    H, W = n_rows * step + overlap, n_cols * step + overlap
    acc, cnt = np.zeros((H, W)), np.zeros((H, W))
    for idx in range(n_rows * n_cols):
        row, col = divmod(idx, n_cols)
        top = margin if row != 0 else 0
        bot = margin if row != n_rows - 1 else 0
        lef = margin if col != 0 else 0
        rig = margin if col != n_cols - 1 else 0
        r0, c0 = row * step, col * step
        acc[r0 + top:r0 + size - bot, c0 + lef:c0 + size - rig] += batched[idx, top:batched.shape[1] - bot, lef:batched.shape[2] - rig]
        cnt[r0 + top:r0 + size - bot, c0 + lef:c0 + size - rig] += 1
    cnt[cnt == 0] = 1
    return acc / cnt


def normalize_preds(hr, hr_hat, pmin: float = 0.1, pmax: float = 99.9):
    r"""Normalizes prediction image intensities to ground truth for fair benchmarking (pssr/util.py:139-191), on the MI355X
    (csrc/metrics.hip, bit-exact with the reference's numpy arithmetic).  ``hr`` / ``hr_hat``: uint8 arrays or tensors of the
    same shape ``[..., H, W]`` -- what ``test_metrics`` and ``predict_images(norm=True)`` pass -- or with image sizes that differ
    (pssr/util.py:176-179: the covariance is then taken against ``skimage.transform.resize(prediction, ground-truth shape)``, here
    its scikit-image >= 0.19 / scipy.ndimage definition evaluated on the device; each output keeps its own size).  Returns uint8
    numpy arrays like the reference."""
    import numpy as np
    from . import ops
    dev = hr_hat.device if isinstance(hr_hat, torch.Tensor) and hr_hat.is_cuda else (hr.device if isinstance(hr, torch.Tensor) and hr.is_cuda else "cuda")
    a, b = (t if isinstance(t, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(t)) for t in (hr, hr_hat))
    if a.dim() != b.dim():
        raise ValueError(f"hr and hr_hat must have the same number of dimensions. Dimension lengths are {tuple(a.shape)} and {tuple(b.shape)} respectively.")
    if a.dtype != torch.uint8 or b.dtype != torch.uint8:
        raise TypeError("normalize_preds on the MI355X takes uint8 images (the output of _pred_array), as test_metrics passes them")
    if a.dim() < 2:
        raise ValueError("images need at least 2 dimensions")
    if a.shape != b.shape:
        n_a, n_b = a.numel() // (a.shape[-1] * a.shape[-2]), b.numel() // (b.shape[-1] * b.shape[-2])
        if n_a != n_b:
            raise ValueError(f"hr and hr_hat must have the same number of images. Received {n_a} and {n_b} images respectively.")
        a2, b2 = ops.normalize_preds_resized_u8(a.to(dev).reshape(-1, *a.shape[-2:]), b.to(dev).reshape(-1, *b.shape[-2:]), pmin, pmax)
        return a2.reshape(a.shape).cpu().numpy(), b2.reshape(b.shape).cpu().numpy()
    shape = a.shape
    a2, b2 = ops.normalize_preds_u8(a.to(dev).reshape(-1, *shape[-2:]), b.to(dev).reshape(-1, *shape[-2:]), pmin, pmax)
    return a2.reshape(shape).cpu().numpy(), b2.reshape(shape).cpu().numpy()


def _sort_tiles(name: str):
    """Sort key of tile names ``{sheet}_{tile}_{slice}[.ext]`` (pssr/util.py:110-114): slice first, then tile."""
    if "." not in name:
        name += "."
    parts = name.replace(".", "_").split("_")
    return int(parts[-2]), int(parts[-3])


def reassemble_sheets(pred_path, lr_path, lr_scale: int, overlap: int = 0, margin: int = 0, out_dir: str = "sheets"):
    r"""Reassembles image sheets from the tiles predicted for a sliding dataset (pssr/util.py:54-108): same arguments, file naming
    and return value.  ``pred_path`` is a directory of tile images or the dict returned by ``predict_images``; the overlap-averaged
    stitching (with ``margin`` trimming) runs on the MI355X (``pssr_patch_tiles_u8``, bit-exact with ``_patch_images`` + the
    uint8 cast).  Multi-frame sheets are written through Pillow (the reference uses tifffile)."""
    import glob
    import os
    import numpy as np
    from PIL import Image
    from . import ops
    if margin > overlap:
        raise ValueError(f"The value of margin cannot be greater than overlap. Given {margin} and {overlap} respectively.")
    sheet_files = glob.glob(f"{lr_path}/*.tif", recursive=True)
    if len(sheet_files) == 0:
        raise FileExistsError("No files exist in lr_path.")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)

    def frames(img):       # pssr/data.py:640-647 (_frame_channel, mode "L")
        return np.stack([np.asarray(f.convert("L"), dtype=np.uint8) for f in _iter_frames(img)])

    def _iter_frames(img):
        for i in range(getattr(img, "n_frames", 1)):
            img.seek(i)
            yield img

    outs = []
    for sheet in sheet_files:
        stem = sheet.split("/")[-1].split(".")[0]
        if type(pred_path) is dict:
            files = sorted([f for f in pred_path if "_".join(f.split("_")[:-2]) == stem], key=_sort_tiles)
            batched = np.asarray([np.asarray(pred_path[f]).squeeze() for f in files])
        else:
            files = sorted(glob.glob(f"{pred_path}/{stem}*"), key=_sort_tiles)
            batched = np.asarray([frames(Image.open(f)).squeeze() for f in files])
        lr_shape = frames(Image.open(sheet)).shape
        size = batched.shape[1]
        n_rows = (lr_shape[1] * lr_scale - size) // (size - overlap * lr_scale) + 1
        n_cols = (lr_shape[2] * lr_scale - batched.shape[2]) // (batched.shape[2] - overlap * lr_scale) + 1
        per = n_rows * n_cols
        tiles = torch.as_tensor(np.ascontiguousarray(batched.astype(np.uint8))).cuda()
        image = np.stack([ops.patch_tiles_u8(tiles[i * per:(i + 1) * per, None].contiguous(), n_rows, n_cols, overlap * lr_scale, margin)[0].cpu().numpy()
                          for i in range(batched.shape[0] // per)])
        if out_dir:
            ims = [Image.fromarray(f) for f in image]
            ims[0].save(f"{out_dir}/{stem}.tif", save_all=len(ims) > 1, append_images=ims[1:])
        else:
            outs.append(image)
    if out_dir is None:
        return outs
